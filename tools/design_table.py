#!/usr/bin/env python
"""Prints the numbers DESIGN.md 5 / README quote from a bench line:  python tools/design_table.py profiles/<tag>_bench.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
c = d.get('launch_calibration') or {}
a = d['also_measured']
g = lambda k, f='ms_per_step': a.get(k, {}).get(f)
print('headline %.3f ms %.1f img/s | eager %s graph %s | depths %s | chosen %s depth %s' % (
    d['ms_per_step'], d['value'], d.get('ms_per_step_eager'), d.get('ms_per_step_graph'),
    [c.get('eager_pipelined_depth%d_ms' % k) for k in (3, 4, 5)], c.get('chosen'), c.get('pipeline_depth')))
for k in a:
    v = a[k]
    print('%-46s %s ms  %s img/s  | %s' % (k, v.get('ms_per_step', v.get('simple_test_ms_per_img')), v.get('img_per_s'), (v.get('launch') or '')[:60]))
r = d['roofline']; rs = d['roofline_step']
print('roofline dominant %.1f TF frac %.4f ms %.4f | wide %s' % (r['achieved'], r['frac'], r['ms_per_launch'],
      {k: d['roofline_wide_schedule'][k] for k in ('achieved', 'frac', 'ms_per_launch')} if 'roofline_wide_schedule' in d else None))
print('step frac %.4f floor frac %s floor_ms %s' % (rs['frac'], rs.get('per_layer_floor', {}).get('frac'), rs.get('per_layer_floor', {}).get('floor_ms')))
for k in ('roofline_gs_loss', 'roofline_gs_loss_n65536', 'roofline_roi_align', 'roofline_merge_score', 'roofline_merge_score_n65536', 'roofline_iou_assign'):
    v = d.get(k)
    if v:
        print(k, {x: v.get(x) for x in ('frac', 'us_per_launch', 'us_per_call', 'traffic', 'algorithmic_bytes')})
print('gs_head', d.get('gs_head'))
print('cpu_baseline', {k: d['cpu_baseline'].get(k) for k in ('value', 'unit', 'cores', 'kind')})
