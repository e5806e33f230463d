#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only) of the bench's roofline
# layer (FPN P2 output conv) under the one-launch form (variant 4) and under the two-launch wide schedule (variant 7 on the
# whole rounds + variant 4 on the left-over rows) on the current tree.   bash tools/pmc_p2_traffic.sh <tag>
set -u
TAG=${1:-pmc_p2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for WIDE in 0 1; do for C in FETCH_SIZE WRITE_SIZE; do
  BGS_HALO_WIDE=$WIDE timeout -k 3 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/w${WIDE}_$C -o k -- python $R/tools/conv_p2_once.py > $OUT/w${WIDE}_$C.log 2> $OUT/w${WIDE}_$C.err
  echo "wide=$WIDE $C rc=$?"
done; done
python - <<PY
import csv, glob, collections, json
res = {}
for wide in (0, 1):
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        agg = collections.defaultdict(list)
        for f in glob.glob('$OUT/w%d_%s/**/*counter_collection.csv' % (wide, C), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r.get('Kernel_Name', '')
                if 'halo_bfx' in kn and r.get('Counter_Name') == C:
                    name = 'bfx7' if 'bfx7' in kn else 'bfx4'
                    agg[name].append(float(r['Counter_Value']))
        for name, v in agg.items():
            v = v[2:]                                 # (the first launches: cold)
            res['wide=%d %s %s' % (wide, name, C)] = dict(n=len(v), avg_kb=sum(v) / len(v))
            print('wide=%d %-5s %-11s n=%d avg=%.1f KB' % (wide, name, C, len(v), sum(v) / len(v)))
json.dump(res, open('$OUT/pmc_p2_traffic.json', 'w'), indent=1)
for wide in (0, 1):
    tot = 0.0
    for name in ('bfx4', 'bfx7'):
        f, w = res.get('wide=%d %s FETCH_SIZE' % (wide, name)), res.get('wide=%d %s WRITE_SIZE' % (wide, name))
        if f and w:
            tot += 2 * f['avg_kb'] + w['avg_kb']
    print('wide=%d: HBM bytes per layer (2 x FETCH + WRITE, all launches of the layer) = %.1f MB' % (wide, tot * 1024 / 1e6))
PY
find $OUT -name "*.csv" -size +5M -delete
