#!/bin/bash
# Timing-only ablations (round 6) of the 3x3 planes kernel on the roofline layer (FPN P2 3x3): what do the filter
# fragments from L2 (p3abl1: loaded twice per chunk instead of 18 times) and the patch loads (p3abl2: loaded once) cost?
# Results of the ablated arms are WRONG.  Build first:
#   for v in p3abl1 p3abl2 p3abl3; do python -m balancedgroupsoftmax_amd.csrc.build --variant $v; done
for rep in 1 2 3; do
  for v in "" p3abl1 p3abl2 p3abl3; do
    if [ -z "$v" ]; then TAG="P2 planes3 full kernel         " python tools/conv_p2_time.py
    else TAG="P2 planes3 $v" BGS_LIB_PATH=balancedgroupsoftmax_amd/libbgs_$v.so python tools/conv_p2_time.py; fi
  done
done
