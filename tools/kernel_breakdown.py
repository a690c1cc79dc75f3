#!/usr/bin/env python
"""Per-(kernel, role) time breakdown of one DenseASPP161 training step (HIP-event timing). Diagnostic."""
import argparse, importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP
from semantic_segmentation_amd import profiling
from bench import synth_batch
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=4); ap.add_argument('--model', default='DenseASPP161')
ap.add_argument('--height', type=int, default=1024); ap.add_argument('--width', type=int, default=2048); ap.add_argument('--steps', type=int, default=2)
a = ap.parse_args()
cfg = importlib.import_module('cfgs.' + a.model).Model_CFG
dev = torch.device('cuda', 0)
model = denseASPP.DenseASPP(cfg).to(dev).train()
x, labels = synth_batch(a.batch, a.height, a.width, 0, dev)
eng = model.engine(a.batch, a.height, a.width, True)
model.refresh_weights()
model.train_step(x, labels, refresh=False); torch.cuda.synchronize()
table = {}; per = {}
progs = [eng._fwd, eng._fwd_loss, eng._bwd]
for p in progs:
    for t, v in profiling.tag_program_fine(p, table).items():
        q = per.setdefault(t, [0.0, 0.0, 0]); q[0] += v[0]; q[1] += v[1]; q[2] += v[2]
for _ in range(a.steps): model.train_step(x, labels, refresh=False)
torch.cuda.synchronize()
nt = len(table) + 1; ms = [0.0]*nt; cnt = [0]*nt
for p in progs:
    m, c = p.read_timing(nt); ms = [u+v for u, v in zip(ms, m)]; cnt = [u+v for u, v in zip(cnt, c)]
rows = []
for key, t in table.items():
    fl, by, n = per[t]
    rows.append((ms[t]/a.steps, key, n, 1e3*ms[t]/max(cnt[t],1), fl*a.steps/(ms[t]*1e-3)/1e12 if fl else 0, by*a.steps/(ms[t]*1e-3)/1e9 if by else 0))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('%-48s %6s %10s %10s %9s %9s' % ('kernel role', 'n', 'ms/step', 'avg us', 'TFLOP/s', 'GB/s'))
for r in rows: print('%-48s %6d %10.3f %10.1f %9.1f %9.1f' % (r[1], r[2], r[0], r[3], r[4], r[5]))
print('total tagged ms/step: %.2f' % tot)
