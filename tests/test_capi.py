"""CPU: the C-ABI library loads and exports every symbol include/bgs.h declares."""
import os
import re

import pytest

from balancedgroupsoftmax_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(names=('bgs.h', 'bgs_tuning.h')):
    """Every function include/*.h declares: the drop-in boundary (bgs.h) and the tuning / census hooks
    (bgs_tuning.h)."""
    out = set()
    for name in names:
        text = open(os.path.join(ROOT, 'include', name)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        out |= set(re.findall(r'\b(bgs_[a-z0-9_]+)\s*\(', text))
    return sorted(out)


def test_header_and_binding_agree():
    assert header_functions() == sorted(capi.SIGNATURES.keys())


def test_library_exports_all_symbols():
    lib = capi.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.bgs_version() >= 100
    assert lib.bgs_error_string(0) == b'ok'
    assert b'unsupported' in lib.bgs_error_string(2)


@pytest.mark.parametrize('variant', ['', 'nodpp'])
def test_variant_files_present(variant):
    from balancedgroupsoftmax_amd.csrc import build as B
    assert os.path.exists(B.lib_path(variant)), 'run python -m balancedgroupsoftmax_amd.csrc.build --all'


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    lib = capi.load()
    assert lib.bgs_gs_loss_fwd_bwd(None, None, None, None, None, 4, 3, 13, None, None, None,
                                   None) == 1
    assert lib.bgs_gs_loss_fwd_bwd(None, None, None, None, None, 4, 99, 13, None, None, None,
                                   None) in (1, 2)
    import ctypes
    import numpy as np
    bad = np.array([[0, 2], [2, 20]], dtype=np.int64)        # bin 1 runs past W=13
    ws = ctypes.create_string_buffer(16)
    assert lib.bgs_gs_loss_fwd_bwd(None, None, bad.ctypes.data_as(ctypes.c_void_p), None, None,
                                   0, 2, 13, None, None, ws, None) == 1
    assert lib.bgs_gs_merge_score(None, None, None, -1, 10, 3, 13, None, None) == 1
    assert lib.bgs_gs_prepare(None, None, None, 0, None, 4, 10, 3, 8.0, 1, None, None, None, None, None) == 1
    assert lib.bgs_bbox_smooth_l1_fwd_bwd(None, None, None, None, 4, 10, 0.0, 4.0, 1.0, None,
                                          None, None, None) == 1
    assert lib.bgs_gs_loss_workspace_bytes(1024, 5) >= 1024 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, '_LIB', None)
    monkeypatch.setenv('BGS_LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(capi.BgsLibraryError):
        capi.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from balancedgroupsoftmax_amd import functional as BF
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        BF.group_softmax_loss(torch.zeros(2, 8), torch.zeros((2, 2), dtype=torch.int32),
                              torch.tensor([[0, 2], [2, 6]]))


def test_head_kernel_variant_rule_is_a_host_decision():
    """``bgs_gs_head_variant_used(N)``: which fused GroupSoftmax head kernel a launch with N rows takes — rows in
    parallel behind one prologue while the per-row partials fit the workspace (N <= 2048), with direct gradient
    stores while the launch is latency-bound (N <= 1024), one row per workgroup with bit planes beyond; an explicit
    variant is honoured where it can run and the multi-row ones fall back beyond 2048 rows; < 0 restores the
    rule.  Pure host logic: no GPU needed."""
    lib = capi.load()
    try:
        lib.bgs_gs_head_variant(-1)
        if os.environ.get('BGS_GS_HEAD_VARIANT') is None:
            assert [lib.bgs_gs_head_variant_used(n) for n in (1, 512, 1023, 1024, 1025, 2048, 2049, 4096)] == \
                [4, 4, 4, 5, 3, 3, 1, 1]
        for v in range(6):
            lib.bgs_gs_head_variant(v)
            assert lib.bgs_gs_head_variant_used(1024) == v
            assert lib.bgs_gs_head_variant_used(4096) == (v if v < 2 else 1)
        lib.bgs_gs_head_variant(17)                   # unknown: the plain kernel
        assert lib.bgs_gs_head_variant_used(1024) == 0
    finally:
        lib.bgs_gs_head_variant(-1)


def test_pointer_parameters_are_void_pointers_so_plain_integer_addresses_convert():
    """``capi.ptr`` / ``capi.current_stream`` hand ctypes plain integers (``tensor.data_ptr()``, the raw stream handle)
    and ``None`` for NULL — no ``c_void_p`` object per argument on the launch path.  That is only valid while every
    pointer-like parameter in ``capi.SIGNATURES`` is declared ``c_void_p``: a typed ``POINTER(...)`` would refuse an int."""
    import ctypes
    import torch
    scalars = (ctypes.c_int, ctypes.c_uint, ctypes.c_long, ctypes.c_ulong, ctypes.c_longlong, ctypes.c_ulonglong,
               ctypes.c_size_t, ctypes.c_float, ctypes.c_double, ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint32,
               ctypes.c_int32, ctypes.c_bool)
    for name, (_res, args) in capi.SIGNATURES.items():
        for a in args:
            assert a is ctypes.c_void_p or a in scalars, (name, a)
    t = torch.arange(6, dtype=torch.float32)
    assert capi.ptr(None) is None and capi.ptr(t) == t.data_ptr() and isinstance(capi.ptr(t), int)
    ctypes.c_void_p.from_param(capi.ptr(t))                  # (what a c_void_p argtype does with it: accepted)
    ctypes.c_void_p.from_param(None)
