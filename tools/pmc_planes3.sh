#!/bin/bash
# PMC passes of the 3x3 planes kernel (conv3x3_planes_bfx_kernel<2>) on the bench's roofline layer (FPN P2 output conv:
# 2 x 200 x 336, 3x3, 256 -> 256; tools/conv_p2_once.py under the default dispatch): HBM-side traffic (FETCH_SIZE /
# WRITE_SIZE), matrix-pipe busy cycles + clock, LDS conflicts, TA / TD.  Separate rocprofv3 --pmc passes, kernel-trace
# only (MI355X_MICROARCH.md; gpurun refuses pmc + other trace domains).   bash tools/pmc_planes3.sh <tag>
set -u
TAG=${1:-pmc_planes3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -k 3 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o k -- python $R/tools/conv_p2_once.py > $OUT/$name.log 2> $OUT/$name.err
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
run ta TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections, json
res = {}
for d in ('fetch', 'write', 'sq', 'ta'):
    agg = collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_planes' in r.get('Kernel_Name', ''):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    ds = []
    for f in glob.glob('$OUT/%s/**/*kernel_trace.csv' % d, recursive=True):
        ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'conv3x3_planes' in r['Kernel_Name']]
    for k, v in sorted(agg.items()):
        v = v[2:]                                   # (the first launches: cold)
        res[k if d != 'ta' or k != 'GRBM_GUI_ACTIVE' else 'GRBM_GUI_ACTIVE_ta'] = sum(v) / len(v)
        print('%-6s %-28s n=%d avg=%.6g' % (d, k, len(v), sum(v) / len(v)))
    if ds:
        ds = ds[2:]
        res['duration_us_' + d] = sum(ds) / len(ds) / 1e3
        print('%-6s duration avg %.1f us (n=%d)' % (d, sum(ds) / len(ds) / 1e3, len(ds)))
if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
    hbm = (2 * res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024
    print('HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KB units): %.1f MB = %.2f x the algorithmic 277.6 MB' % (hbm / 1e6, hbm / 277610496.0))
    res['traffic_bytes_per_launch'] = hbm
if 'GRBM_GUI_ACTIVE' in res and 'duration_us_sq' in res:
    cyc = res['GRBM_GUI_ACTIVE'] / 8
    print('clock %.2f GHz; matrix pipe busy %.3f of the cycles' % (cyc / res['duration_us_sq'] / 1e3, res['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)))
    res['effective_clock_ghz'] = cyc / res['duration_us_sq'] / 1e3
    res['matrix_pipe_busy'] = res['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
json.dump(res, open('$OUT/pmc_planes3.json', 'w'), indent=1)
PY
find $OUT -name "*.csv" -size +5M -delete
