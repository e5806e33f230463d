#!/usr/bin/env python
"""TEST INFRASTRUCTURE: builds the pieces of the REAL reference that compile from their own
source files, in place, into ``oracle/_ref/`` (git-ignored, but it travels to the GPU box).

    python oracle/build_ref.py

* ``nms_cpu_ref.so``  <- /root/reference/mmdet/ops/nms/src/nms_cpu.cpp (ATen C++, one file;
  compiled with g++ through torch.utils.cpp_extension; ``AT_ASSERTM`` — removed from current
  torch — is mapped to ``TORCH_CHECK`` on the command line, the source is untouched).

Not buildable here (documented in DESIGN.md): ``nms_kernel.cu`` / ``roi_align_kernel.cu``
(CUDA + THC, no nvcc) — those two have numpy restatements in ``oracle/det_oracle.py``.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('BGS_REFERENCE_ROOT', '/root/reference')
OUT = os.path.join(HERE, '_ref')


def build_nms_cpu():
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(REF, 'mmdet/ops/nms/src/nms_cpu.cpp')
    return load(name='nms_cpu_ref', sources=[src], build_directory=OUT, verbose=False,
                extra_cflags=['-O2', '-DAT_ASSERTM=TORCH_CHECK'])


def load_nms_cpu():
    """Import the prebuilt module (GPU box: /root/reference is absent, the .so is present)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch symbols)
    path = os.path.join(OUT, 'nms_cpu_ref.so')
    if not os.path.exists(path):
        if os.path.isdir(REF):
            return build_nms_cpu()
        return None
    spec = importlib.util.spec_from_file_location('nms_cpu_ref', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference tree not found at %s' % REF)
    m = build_nms_cpu()
    print('built', os.path.join(OUT, 'nms_cpu_ref.so'), m)
