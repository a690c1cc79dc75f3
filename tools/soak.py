#!/usr/bin/env python
"""Soak run (diagnostic): N optimizer steps of DenseASPP161 at 1024x2048 on a small set of fixed synthetic batches, reference
optimizer settings (Adam, lr 1e-4 here 3e-4 to see movement, coupled L2 1e-4).  Reports the loss every few steps, checks that
every loss and the parameter arena stay finite, that the allocator's footprint does not grow, and that the loss falls."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib, denseASPP
from semantic_segmentation_amd.optim import FusedAdam
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=300); ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--model', default='DenseASPP161'); ap.add_argument('--nbatches', type=int, default=4); ap.add_argument('--lr', type=float, default=3e-4)
a = ap.parse_args()
cfg = importlib.import_module('cfgs.' + a.model).Model_CFG
torch.manual_seed(0)
model = denseASPP.DenseASPP(cfg, n_class=19, output_stride=8).cuda().train()
opt = FusedAdam(model, lr=a.lr, weight_decay=1e-4)
H, W = 1024, 2048
g = torch.Generator(device='cuda'); g.manual_seed(1)
data = []
for i in range(a.nbatches):
    # piecewise-constant label maps with a learnable relation to the image: class = quantised local brightness of channel 0
    x = torch.randn(a.batch, 3, H // 64, W // 64, device='cuda', generator=g)
    x = torch.nn.functional.interpolate(x, size=(H, W), mode='bilinear', align_corners=False) + 0.1 * torch.randn(a.batch, 3, H, W, device='cuda', generator=g)
    lab = ((x[:, 0] + 2.0) * 4.75).clamp(0, 18).long()
    lab[:, :8, :] = 255
    data.append((x.contiguous(), lab))
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
losses = []; mem0 = None; t0 = time.time()
for s in range(a.steps):
    x, lab = data[s % a.nbatches]
    loss = model.train_step(x, lab, generator=gen)
    opt.step()
    if s % 20 == 0 or s == a.steps - 1:
        l = float(loss); losses.append(l)
        mem = torch.cuda.memory_allocated()
        if mem0 is None and s >= 20: mem0 = mem
        print('step %4d loss %.4f  allocated %.2f GB  %.1f s' % (s, l, mem / 1e9, time.time() - t0), flush=True)
        assert l == l and abs(l) < 1e4, 'loss not finite'
torch.cuda.synchronize()
assert torch.isfinite(model._flat).all(), 'non-finite parameter'
assert torch.cuda.memory_allocated() <= (mem0 or 0) * 1.001 + 1e6, 'allocator footprint grew'
print('first %.4f last %.4f min %.4f; %d steps in %.1f s; finite parameters; footprint constant' % (losses[0], losses[-1], min(losses), a.steps, time.time() - t0))
assert losses[-1] < 0.7 * losses[0], 'loss did not fall'
