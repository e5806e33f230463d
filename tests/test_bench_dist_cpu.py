"""bench.py's N > 1 orchestration (bench_dist.run) on CPU: gloo, world size 8, a stand-in step with the interface of
bench.DetectorStep, every diagnostic forced to fail — and one rank forced to hang — through the test hooks.  What must
hold in every scenario: exactly ONE line is emitted on rank 0, every rank leaves with code 0, the
headline measurement is in the line whenever the timed region ran, a failed diagnostic is ``{"error": ...}`` in its own
field and costs no other field, and the whole scenario ends within its time budget.
(the reference: tools/dist_train.sh:8-9, mmdet/core/utils/dist_utils.py:9-58, mmdet/apis/train.py:143-205)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import bench_dist_dryrun as dry  # noqa: E402

WORLD = 8


def _run(env, port_off):
    lines, wall, ok = dry.run_scenario(env, WORLD, port=37000 + port_off + os.getpid() % 1500)
    assert ok, 'a rank left with a non-zero exit code'
    assert len(lines) == 1, lines
    return lines[0], wall


def _headline_ok(f):
    assert f['ms_per_step'] > 0 and len(f['rank_ms']) == WORLD
    assert f['launch_calibration']['chosen'] in ('forks_on', 'pipelined')


def test_clean_run_measures_everything():
    f, wall = _run({}, 0)
    _headline_ok(f)
    d = f['diagnostics']
    assert d['grad_exchange_check']['checked'] and d['grad_exchange_check']['ok']
    assert d['grad_exchange_check']['ranks_see_different_data']
    assert isinstance(d['allreduce_us'], float) and d['allreduce_us'] > 0
    assert d['n1_same_invocation']['ms_per_step'] > 0 and d['weak_scaling_eff'] > 0
    assert set(f['launch_calibration']['ms_by_rank']) == {'forks_on', 'pipelined'}      # the default arms only
    assert wall < 60


def test_full_calibration_is_opt_in():
    f, _ = _run({'BGS_BENCH_DIST_CALIB': 'full'}, 10)
    assert set(f['launch_calibration']['ms_by_rank']) == {'forks_on', 'forks_off', 'pipelined', 'pipelined_depth5'}


@pytest.mark.parametrize('env,field', [
    ({'BGS_BENCH_FAIL': 'grad_exchange_check@3'}, 'grad_exchange_check'),
    ({'BGS_BENCH_FAIL': 'grad_exchange_check'}, 'grad_exchange_check'),
    ({'BGS_BENCH_FAIL': 'allreduce_us@7'}, 'allreduce_us'),
    ({'BGS_BENCH_FAIL': 'allreduce_us'}, 'allreduce_us'),
    ({'BGS_BENCH_FAIL': 'n1_reference@0'}, 'n1_same_invocation'),
])
def test_a_failing_diagnostic_costs_only_its_own_field(env, field):
    f, wall = _run(env, 20 + hash(str(env)) % 200)
    _headline_ok(f)
    d = f['diagnostics']
    assert 'error' in d[field], d[field]
    others = {'grad_exchange_check', 'allreduce_us', 'n1_same_invocation'} - {field}
    for o in others:
        v = d[o]
        assert not (isinstance(v, dict) and 'error' in v), (o, v)
    assert wall < 60


def test_failure_before_the_calibration_skips_it_on_every_rank():
    f, _ = _run({'BGS_BENCH_FAIL': 'calibration@5'}, 300)
    assert f['ms_per_step'] > 0 and f['pipeline_depth'] == 0
    assert 'skipped' in f['launch_calibration']['note']
    assert f['diagnostics']['grad_exchange_check']['ok']


def test_every_diagnostic_failing_still_gives_one_line():
    f, _ = _run({'BGS_BENCH_FAIL': 'grad_exchange_check@1,allreduce_us@2,n1_reference@0'}, 320)
    _headline_ok(f)
    d = f['diagnostics']
    assert all('error' in d[k] for k in ('grad_exchange_check', 'allreduce_us', 'n1_same_invocation'))


def test_a_hung_rank_costs_the_diagnostics_budget_not_the_line():
    f, wall = _run({'BGS_BENCH_HANG': 'grad_exchange_check@2', 'BGS_BENCH_DIAG_SECONDS': '4'}, 340)
    _headline_ok(f)
    assert 'grad_exchange_check unfinished' in f['watchdog']
    assert wall < 40


def test_a_hang_before_the_timed_region_ends_with_an_error_line():
    f, wall = _run({'BGS_BENCH_HANG': 'calibration@6', 'BGS_BENCH_WALL_SECONDS': '4'}, 360)
    assert 'ms_per_step' not in f and 'wall budget' in f['error']
    assert wall < 40
