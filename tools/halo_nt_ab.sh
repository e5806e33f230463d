#!/bin/bash
# A/B (round 6): non-temporal hints on the activation side of the 3x3 halo kernels (BGS_HALO_NT bit 0 = patch loads,
# bit 1 = output stores) on the roofline layer (FPN P2 3x3), one-launch and two-launch (wide) schedules, interleaved.
for rep in 1 2; do
  for nt in 0 1 2 3; do
    TAG="P2 variant4 nt=$nt" BGS_HALO_NT=$nt python tools/conv_p2_time.py
    TAG="P2 wide     nt=$nt" BGS_HALO_NT=$nt BGS_HALO_WIDE=1 python tools/conv_p2_time.py
  done
done
