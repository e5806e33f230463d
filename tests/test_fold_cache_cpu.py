"""CPU: the version-keyed cache of folded conv + BN weights (backbone._FoldCache / cached_fold) — the check sits on the
launch path of every frozen layer, so it is cheap, and it must still see every way a weight can change."""
import torch
from torch import nn

from balancedgroupsoftmax_amd import backbone as B


def _pair():
    torch.manual_seed(0)
    conv = nn.Conv2d(8, 16, 1, bias=False)
    bn = nn.BatchNorm2d(16).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
        bn.weight.uniform_(0.5, 2)
        bn.bias.uniform_(-1, 1)
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad = False
    return conv, bn


def test_cached_fold_is_reused_and_sees_every_kind_of_update():
    conv, bn = _pair()
    calls = []
    real = B._fold_conv_bn

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)

    B._fold_conv_bn = spy
    try:
        w0, b0 = B.cached_fold(conv, bn)
        w1, _ = B.cached_fold(conv, bn)
        assert len(calls) == 1 and w1 is w0                       # reused: no re-listing, no rebuild
        ref_w = conv.weight * (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(-1, 1, 1, 1)
        assert torch.allclose(w0.permute(0, 3, 1, 2), ref_w, atol=1e-6)
        with torch.no_grad():
            conv.weight.mul_(2.0)                                 # in-place update: version counter
        w2, _ = B.cached_fold(conv, bn)
        assert len(calls) == 2 and torch.allclose(w2, 2 * w0, atol=1e-6)
        with torch.no_grad():
            bn.running_var.add_(1.0)                              # a buffer moves
        B.cached_fold(conv, bn)
        assert len(calls) == 3
        conv.weight = nn.Parameter(conv.weight.detach().clone() * 0.5, requires_grad=False)   # re-registered object
        w4, _ = B.cached_fold(conv, bn)
        assert len(calls) == 4 and torch.allclose(w4, w0 * (bn.weight / torch.sqrt(bn.running_var + bn.eps) /
                                                             (bn.weight / torch.sqrt(bn.running_var - 1.0 + bn.eps))
                                                             ).view(-1, 1, 1, 1), atol=1e-5)
        conv.weight.data = conv.weight.data.clone()               # new storage, same object, same version
        B.cached_fold(conv, bn)
        assert len(calls) == 5
        B.cached_fold(conv, bn)
        assert len(calls) == 5
        # a trained parameter: folded on every call while grad is enabled, cached under no_grad
        conv.weight.requires_grad = True
        B.cached_fold(conv, bn)
        B.cached_fold(conv, bn)
        assert len(calls) == 7
        with torch.no_grad():                                     # nothing changed since the last cached fold
            B.cached_fold(conv, bn)
            B.cached_fold(conv, bn)
        assert len(calls) == 7
    finally:
        B._fold_conv_bn = real


def test_fold_cache_relists_when_the_set_of_registered_names_changes():
    conv, _ = _pair()
    cache = B._FoldCache()
    n = []
    build = lambda: n.append(1) or len(n)
    assert cache.get(conv, build) == 1 and cache.get(conv, build) == 1
    conv.register_buffer('extra', torch.zeros(3))                 # one more registered name
    assert cache.get(conv, build) == 2 and cache.get(conv, build) == 2
    del conv._buffers['extra']                                    # and one less
    assert cache.get(conv, build) == 3
    other, _ = _pair()
    assert cache.get(other, build) == 4 and cache.get((other,), build) == 4     # another module; tuple form


def test_fold_cache_sees_a_sub_module_replaced_under_the_same_name():
    """``m[0] = new_conv`` keeps the number of registered names; the cached slots pointed at the OLD module's
    ``_parameters`` and a stale fold came back.  The identities of the registered sub-modules are part of the check."""
    m = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1, bias=False), torch.nn.BatchNorm2d(4))
    for p in m.parameters():
        p.requires_grad = False
    cache = B._FoldCache()
    build = lambda: float(m[0].weight.sum())                      # noqa: E731  (what a fold would read)
    first = cache.get(m, build)
    assert cache.get(m, build) == first
    new_conv = torch.nn.Conv2d(4, 4, 1, bias=False)
    for p in new_conv.parameters():
        p.requires_grad = False
    m[0] = new_conv                                               # same name '0', same count
    second = cache.get(m, build)
    assert second == float(new_conv.weight.sum()) and second != first
    assert cache.get(m, build) == second
    blk = B.Bottleneck(8, 2)                                      # the product's user: Bottleneck.folded()
    for p in blk.parameters():
        p.requires_grad = False
    blk.eval()
    w_old = blk.folded()['c1'][0].clone()
    blk.conv1 = torch.nn.Conv2d(8, 2, 1, bias=False).requires_grad_(False)
    assert not torch.equal(blk.folded()['c1'][0], w_old)
