"""Launches the GroupSoftmax kernels the bench line prices (fused head kernel at N = 1024, row kernel at N = 65,536) a
few times — the target of tools/pmc_gs.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
small = bench.make_inputs(1024, seed=1000, dev=dev)
big = bench.make_inputs(65536, seed=7, dev=dev)
print(bench.kernel_roofline(small, 1024, iters=10, kernel='fused')['kernel'])
print(bench.kernel_roofline(big, 65536, iters=6, kernel='rowwave')['kernel'])
