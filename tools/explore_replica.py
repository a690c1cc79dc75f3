#!/usr/bin/env python
"""Run-to-run and replicated-batch differences of DenseASPP201 training gradients at 1024x2048 (random initialisation)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import denseASPP
from cfgs import DenseASPP201
from e2e_common import grad_report

def batch(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g).clamp_(-2.2, 2.7)
    labels = torch.randint(0, 19, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return x.cuda(), labels.cuda()

torch.manual_seed(0)
m = denseASPP.DenseASPP(DenseASPP201.Model_CFG, n_class=19, output_stride=8).cuda().train()
state = {k: v.clone() for k, v in m.state_dict().items()}
H, W, B0 = 1024, 2048, 10
x, labels = batch(B0, H, W, 51)
sp = m.spec
g = torch.Generator().manual_seed(3)
masks = {a.prefix: ((torch.rand(B0, sp.d1, generator=g) >= sp.dropout0).float() / (1 - sp.dropout0)).cuda() for a in sp.aspp}
masks['classification'] = ((torch.rand(B0, sp.C_aspp, generator=g) >= sp.dropout1).float() / (1 - sp.dropout1)).cuda()

def run(rep):
    m._engines.clear(); torch.cuda.empty_cache()
    m.load_state_dict(state)
    loss = float(m.train_step(x.repeat(rep, 1, 1, 1), labels.repeat(rep, 1, 1), dropout_masks={k: v.repeat(rep, 1) for k, v in masks.items()}))
    torch.cuda.synchronize()
    return loss, {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}

def cmp(tag, a, b):
    def group(prefix):
        num = sum(float((a[k] - b[k]).pow(2).sum()) for k in a if k.startswith(prefix))
        return (num / sum(float(a[k].pow(2).sum()) for k in a if k.startswith(prefix))) ** 0.5
    rel, cos, worst = grad_report(a, b)
    print(tag, 'rel %.2e cos %.6f' % (rel, cos), {p: '%.1e' % group(p) for p in ('classification', 'ASPP_', 'features.norm5', 'features.transition4', 'features.denseblock4', 'features.denseblock3', 'features.denseblock1')}, flush=True)

l1, g1 = run(1)
l1b, g1b = run(1)
print('loss', l1, l1b)
cmp('B=10 twice      ', g1, g1b)
l2, g2 = run(2)
print('loss x2', l2)
cmp('B=20 (2x) vs 10 ', g1, g2)
l4, g4 = run(4)
print('loss x4', l4)
cmp('B=40 (4x) vs 10 ', g1, g4)
cmp('B=40 vs B=20    ', g2, g4)
