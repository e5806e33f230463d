"""CPU: the numpy oracle must reproduce what the EXECUTED reference class produced
(fixtures from tests/golden/make_golden.py).  This is the pin of the oracle."""
import numpy as np
import pytest

from oracle import gs_oracle
from tests.golden_util import case_names, case_setup, golden

C = 1231


@pytest.mark.parametrize('name', case_names())
def test_remap_and_sampling_bit_exact(name):
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    np.random.seed(case['seed'])
    bl, w, avg = gs_oracle.remap_and_sample(batch['labels'], l2b, case.get('ratio', 8.0), cls_w)
    g = golden()
    np.testing.assert_array_equal(w, g.get(name, 'weights'))
    np.testing.assert_array_equal(avg, g.get(name, 'avg'))
    # label2binlabel gather is an integer op: verify against the definition
    for i in range(l2b.shape[0]):
        np.testing.assert_array_equal(bl[i], l2b[i][batch['labels']])


@pytest.mark.parametrize('name', case_names())
def test_group_softmax_loss_and_grad(name):
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    g = golden()
    bl = gs_oracle.remap_labels(batch['labels'], l2b)
    w, avg = g.get(name, 'weights'), g.get(name, 'avg')
    losses, dz = gs_oracle.group_softmax_loss(batch['logits'], bl, w, avg, ps)
    np.testing.assert_allclose(losses, g.get(name, 'losses'), rtol=2e-6, atol=2e-6)
    rows = g.get(name, 'grad_rows')
    np.testing.assert_allclose(dz[rows], g.get(name, 'grad_sub'), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(np.abs(dz).sum(1), g.get(name, 'grad_rowl1'), rtol=1e-5, atol=1e-7)
    # fp32 flavour of the oracle stays within the 1e-4 parity budget too
    l32, d32 = gs_oracle.group_softmax_loss(batch['logits'], bl, w, avg, ps, dtype=np.float32)
    np.testing.assert_allclose(l32, g.get(name, 'losses'), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d32[rows], g.get(name, 'grad_sub'), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('name', case_names())
def test_bbox_loss_and_grad(name):
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    if case.get('no_bbox'):
        pytest.skip('reference asserts on an all-background batch (smooth_l1_loss.py:11)')
    g = golden()
    loss, grad = gs_oracle.smooth_l1_bbox_loss(
        batch['bbox_pred'], batch['labels'], batch['bbox_targets'], batch['bbox_weights'],
        C, reg_class_agnostic=bool(case.get('agnostic')))
    np.testing.assert_allclose(loss, g.get(name, 'loss_bbox'), rtol=2e-6, atol=1e-7)
    idx = g.get(name, 'gbbox_idx')
    flat = grad.reshape(-1)
    np.testing.assert_allclose(flat[idx], g.get(name, 'gbbox_val'), rtol=1e-5, atol=1e-9)
    mask = np.ones(flat.shape[0], dtype=bool)
    mask[idx] = False
    assert np.abs(flat[mask]).max(initial=0.0) < 1e-12


@pytest.mark.parametrize('name', [n for n in case_names() if golden().has(n, 'merge_sub')])
def test_merge_score(name):
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    g = golden()
    ms = gs_oracle.merge_score(batch['logits'] * np.float32(2.0), ps, fg_splits, C)
    rows = g.get(name, 'grad_rows')
    np.testing.assert_allclose(ms[rows], g.get(name, 'merge_sub'), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(ms.sum(1), g.get(name, 'merge_rowsum'), rtol=1e-5)


def test_reference_observed_values_cfg1():
    """Bin widths of the synthetic 5-bin split (SURVEY.md §8d rule; the survey used a different draw: 286/265)."""
    case, l2b, ps, fg_splits, cls_w, batch = case_setup('n512_cfg1')
    assert ps[:, 1].tolist() == [2, 285, 312, 266, 371]
    assert int(ps[-1].sum()) == 1236
    # every fg class lives in exactly one fg bin
    assert ((l2b[1:, 1:] > 0).sum(0) == 1).all()


@pytest.mark.parametrize('name', ['n512_cfg1', 'n1024_cfg2', 'n256_ratio2', 'n96_3bins'])
def test_torch_port_matches_reference_fixtures(name):
    """The torch-CPU port timed as cpu_baseline reproduces the executed reference."""
    import torch
    from oracle import gs_torch_port
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    g = golden()
    np.random.seed(case['seed'])
    losses, grad = gs_torch_port.gs_loss_fwd_bwd(
        torch.from_numpy(batch['logits']), torch.from_numpy(batch['labels']),
        torch.from_numpy(l2b), torch.from_numpy(ps), case.get('ratio', 8.0))
    np.testing.assert_allclose(losses.numpy(), g.get(name, 'losses'), rtol=1e-6, atol=1e-6)
    rows = g.get(name, 'grad_rows')
    np.testing.assert_allclose(grad.numpy()[rows], g.get(name, 'grad_sub'), rtol=1e-6, atol=1e-9)
