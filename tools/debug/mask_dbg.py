import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import balancedgroupsoftmax_amd as bgs
from oracle import mask_oracle
from tests.golden import make_golden_mask
from tests.test_gpu_mask import _head, GOLD
z = np.load(GOLD)
name = 'p6_c1231'
case = [c for c in json.loads(bytes(z['__cases__']).decode()) if c['name'] == name][0]
head = _head(case['C'])
with torch.no_grad():
    mask_oracle.fill_mask_head(head.state_dict(), case['seed'] + 1000)
head.to('cuda:0')
feats, labels, targets = make_golden_mask.case_inputs(case)
x = torch.from_numpy(feats).permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
lab = torch.from_numpy(labels).cuda()
f = head.features(x, nhwc=True)
loss = head.loss_from_features(f, torch.from_numpy(targets).cuda(), lab)['loss_mask']
loss.sum().backward()
def rep(nm, a, b):
    rel = np.abs(a - b) / max(np.abs(b).max(), 1e-12)
    print(nm, 'max rel', rel.max(), 'frac<2e-4', (rel < 2e-4).mean(), 'n>2e-4', int((rel >= 2e-4).sum()), 'of', rel.size)
rep('dx', x.grad.permute(0, 3, 1, 2)[:, :, ::5, ::3].cpu().numpy(), z[name + '/dx'])
rep('dw', head.conv_logits.weight.grad.view(case['C'], 256)[lab].cpu().numpy(), z[name + '/dw_rows'].reshape(case['P'], 256))
rep('dconv0', head.convs[0].conv.weight.grad[::16, ::16].cpu().numpy(), z[name + '/dconv0_w'])
rep('dupb', head.upsample.bias.grad.cpu().numpy(), z[name + '/dup_b'])
