"""Builds libbgs.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m balancedgroupsoftmax_amd.csrc.build [--variant nodpp] [--force]

The .so lands next to the package (balancedgroupsoftmax_amd/libbgs.so) so that it travels to
the GPU box with the repo snapshot.  hipcc cross-compiles without a GPU.
"""
import argparse
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, 'include')

VARIANTS = {
    '': dict(name='libbgs.so', flags=[]),
    'nodpp': dict(name='libbgs_nodpp.so', flags=['-DBGS_NO_DPP']),
}
# built on request only (python -m ...build --variant ablate): timing-only instantiations of the
# conv loops with one component removed (tools/ablate.py); never loaded by the product
EXTRA_VARIANTS = {
    'ablate': dict(name='libbgs_ablate.so', flags=['-DBGS_ABLATE']),
    # the operand split's residual by shift / mask + subtraction instead of v_dot2c_f32_bf16 (csrc/bfx_split.h): the A/B
    # arm of tools/split_ab.sh, loaded through BGS_LIB_PATH
    'splitsub': dict(name='libbgs_splitsub.so', flags=['-DBGS_SPLIT_SUB']),
    # cache-policy bits on the activation loads / output stores of the 3x3 planes kernel (A/B, tools/planes3_nt_ab.sh):
    # aux 2 = non-temporal
    'p3nta': dict(name='libbgs_p3nta.so', flags=['-DBGS_P3_A_AUX=2']),
    'p3nty': dict(name='libbgs_p3nty.so', flags=['-DBGS_P3_Y_AUX=2']),
    'p3ntay': dict(name='libbgs_p3ntay.so', flags=['-DBGS_P3_A_AUX=2', '-DBGS_P3_Y_AUX=2']),
    # timing-only ablations of the 3x3 planes kernel: 1 = filter fragments from L2 twice per chunk instead of 18 times,
    # 2 = the patch loaded once, 3 = both (results are wrong; tools/planes3_ablate.sh)
    'p3prio': dict(name='libbgs_p3prio.so', flags=['-DBGS_P3_PRIO=1']),
    'p3abl1': dict(name='libbgs_p3abl1.so', flags=['-DBGS_P3_ABL=1']),
    'p3abl2': dict(name='libbgs_p3abl2.so', flags=['-DBGS_P3_ABL=2']),
    'p3abl3': dict(name='libbgs_p3abl3.so', flags=['-DBGS_P3_ABL=3']),
}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def sources():
    return sorted(glob.glob(os.path.join(HERE, '*.hip')))


def _digest(flags):
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(HERE, '*.h'))) + \
            sorted(glob.glob(os.path.join(INCLUDE, '*.h'))):
        h.update(f.encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def lib_path(variant=''):
    return os.path.join(PKG, {**VARIANTS, **EXTRA_VARIANTS}[variant]['name'])


def build(variant='', force=False, verbose=True):
    v = {**VARIANTS, **EXTRA_VARIANTS}[variant]
    out = lib_path(variant)
    stamp = out + '.stamp'
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
             '-I' + INCLUDE, '-I' + HERE, '-Wno-unused-result'] + v['flags']
    dig = _digest(flags)
    if not force and os.path.exists(out) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == dig:
                return out
    objs = []
    os.makedirs(os.path.join(PKG, 'build', variant or 'default'), exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(PKG, 'build', variant or 'default',
                           os.path.basename(src).replace('.hip', '.o'))
        cmd = [_hipcc()] + [f for f in flags if f != '-shared'] + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        objs.append(obj)
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    # --no-undefined: a kernel template whose host stub failed to instantiate (seen with an inline-asm operand the
    # HOST pass rejects silently) must fail the build, not the first dlopen on the GPU box
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--no-undefined', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(dig)
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--variant', default='', choices=sorted({**VARIANTS, **EXTRA_VARIANTS}))
    ap.add_argument('--all', action='store_true')
    ap.add_argument('--force', action='store_true')
    a = ap.parse_args()
    for var in (sorted(VARIANTS) if a.all else [a.variant]):
        print(build(var, force=a.force))
