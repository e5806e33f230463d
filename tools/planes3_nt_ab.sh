#!/bin/bash
# A/B (round 6): non-temporal cache policy on the activation loads (p3nta), the output stores (p3nty) or both (p3ntay) of
# the 3x3 planes kernel, on the roofline layer (FPN P2 3x3) — does keeping the streaming side out of L2 leave the filter
# resident (HBM-side traffic is 1.93 x algorithmic)?  Build the variants first:
#   for v in p3nta p3nty p3ntay; do python -m balancedgroupsoftmax_amd.csrc.build --variant $v; done
for rep in 1 2 3; do
  for v in "" p3nta p3nty p3ntay; do
    if [ -z "$v" ]; then TAG="P2 planes3 default   " python tools/conv_p2_time.py
    else TAG="P2 planes3 $v" BGS_LIB_PATH=balancedgroupsoftmax_amd/libbgs_$v.so python tools/conv_p2_time.py; fi
  done
done
