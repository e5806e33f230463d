#!/bin/bash
# SQ / GRBM counters of the halo 3x3 kernels on the two-round map: variant 4, variant 7, and variant 7's timing-only
# ablations (no filter DMA / nothing but MFMAs).  GRBM_GUI_ACTIVE / duration = the clock the chip sustains.
set -u
TAG=${1:-pmc_halo_wide}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for ARM in "0 0" "1 0" "1 4" "1 7"; do
  set -- $ARM
  D=$OUT/w$1_a$2
  timeout -k 3 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $D -o k -- python $R/tools/halo_wide_once.py $1 $2 > $D.log 2> $D.err
  echo "arm $ARM rc=$?"
done
python - <<PY
import csv, glob, collections
for arm in ['w0_a0', 'w1_a0', 'w1_a4', 'w1_a7']:
    cc = glob.glob('$OUT/%s/**/*counter_collection.csv' % arm, recursive=True)
    kt = glob.glob('$OUT/%s/**/*kernel_trace.csv' % arm, recursive=True)
    if not cc or not kt:
        print(arm, 'missing'); continue
    dur = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(kt[0])) if 'halo_bfx' in r['Kernel_Name']]
    dur = dur[2:]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'halo_bfx' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    ns = sum(dur) / len(dur)
    line = '%s: %.1f us' % (arm, ns / 1e3)
    g = agg.get('GRBM_GUI_ACTIVE')
    if g:
        gv = sum(g[2:]) / len(g[2:])
        line += ' | GRBM_GUI_ACTIVE %.4g (sum of XCDs) -> clock %.3f GHz' % (gv, gv / 8 / ns)
        m = agg.get('SQ_VALU_MFMA_BUSY_CYCLES')
        if m:
            mv = sum(m[2:]) / len(m[2:])
            line += ' | MFMA busy %.3f' % (mv / (1024 * gv / 8))
    for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_MFMA'):
        v = agg.get(k)
        if v:
            line += ' | %s %.4g' % (k, sum(v[2:]) / len(v[2:]))
    print(line)
PY
find $OUT -name "*.csv" -size +5M -delete
