#!/usr/bin/env python
"""Folds gpurun_out/<tag>/pmc_hbm_kernels.json (tools/pmc_hbm_kernels.sh) into profiles/pmc_traffic.json — the entries
bench.py's roofline_roi_align / roofline_merge_score* / roofline_iou_assign report as `traffic`.
HBM bytes per launch = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md for 16-byte-per-lane streaming
reads; checked here against the merge kernel at R = 65,536, whose read is exactly R x 4,944 B) + WRITE_SIZE.
python tools/pmc_merge_into_traffic.py gpurun_out/<tag>/pmc_hbm_kernels.json <source note>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, note = sys.argv[1], sys.argv[2]
d = json.load(open(src))
out_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
out = json.load(open(out_path))


def traffic(ent):
    return int(round((2 * ent['FETCH_SIZE']['avg_kb'] + ent['WRITE_SIZE']['avg_kb']) * 1024))


def put(kernel, key, ents, extra=None):
    e = dict(traffic_bytes_per_launch=sum(traffic(x) for x in ents),
             fetch_size_kb=round(sum(x['FETCH_SIZE']['avg_kb'] for x in ents), 1), fetch_correction=2.0,
             write_size_kb=round(sum(x['WRITE_SIZE']['avg_kb'] for x in ents), 1), source=note)
    if extra:
        e.update(extra)
    out.setdefault(kernel, {})[str(key)] = e
    print(kernel, key, e['traffic_bytes_per_launch'])


for k, ent in d.items():
    name, grid = k.split(' grid=')
    grid = int(grid)
    if name.startswith('roi_align_fwd_grid_kernel<1,false>'):
        put('roi_align_fwd_grid_kernel<1,false>', grid // 64 // 49, [ent])
    elif name.startswith('roi_align_nhwc_kernel<2,false,1,false>'):
        put('roi_align_nhwc_kernel<2,false,1,false>', grid // 64 // 49, [ent])
    elif name.startswith('roi_align_nhwc_kernel<2,true,1,false>'):
        put('roi_align_nhwc_kernel<2,true,1,false>', grid // 64 // 49, [ent])
    elif name == 'gs_merge_rowwave_kernel':
        put('gs_merge_rowwave_kernel', 1000 if grid == 256000 else 65536, [ent])
big = [(k, e) for k, e in d.items() if k.startswith('iou_') and int(k.split('grid=')[1]) > 100000]
if len(big) == 2:
    A = 268569
    put('iou_gtmax_kernel+iou_assign_kernel', A, [e for _, e in big])
json.dump(out, open(out_path, 'w'), indent=1)
