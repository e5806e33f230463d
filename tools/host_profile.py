"""cProfile of the launching thread over n pipelined steps (where does the host's time per launch go?).
python tools/host_profile.py [steps=30] [--cascade] [--bf16] [--seq]"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import torch
import bench
from balancedgroupsoftmax_amd import functional as BF

n = int(argv[0]) if argv else 30
flags = argv[1:]
dev = torch.device('cuda', 0)
if '--bf16' in flags:
    BF.set_conv_math('bf16')
x101 = '--cascade' in flags or '--htc' in flags
step = bench.DetectorStep(dev, 0, 1, 2, 3 if x101 else 1, mask='--mask' in flags, cascade='--cascade' in flags,
                          htc='--htc' in flags, conv_math='bf16' if '--bf16' in flags else 'bf16x6')
fn = step if '--seq' in flags else step.pipelined(depth=3 if x101 else 5)
for _ in range(8):
    fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    fn()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
    print('\n'.join(l[:150] for l in s.getvalue().split('\n')[:58]))
