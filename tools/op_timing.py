#!/usr/bin/env python
"""Per-operation time of one DenseASPP training step, every op of the recorded programs, grouped by (kind, role, shape)
(HIP-event timing around each op; diagnostic)."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP
from semantic_segmentation_amd import profiling
from bench import synth_batch
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16); ap.add_argument('--model', default='DenseASPP161')
ap.add_argument('--height', type=int, default=1024); ap.add_argument('--width', type=int, default=2048); ap.add_argument('--steps', type=int, default=2)
ap.add_argument('--top', type=int, default=60)
a = ap.parse_args()
cfg = importlib.import_module('cfgs.' + a.model).Model_CFG
dev = torch.device('cuda', 0)
model = denseASPP.DenseASPP(cfg).to(dev).train()
x, labels = synth_batch(a.batch, a.height, a.width, 0, dev)
eng = model.engine(a.batch, a.height, a.width, True)
model.refresh_weights()
model.train_step(x, labels, refresh=False); torch.cuda.synchronize()
progs = [('fwd', eng._fwd), ('loss', eng._fwd_loss), ('bwd', eng._bwd)]
table = {}; work = {}
def sig(pn, kind, f):
    g = lambda k: f.get(k) if f.get(k) is not None else 0
    if kind in ('conv', 'wgrad'):
        M = f['n_img'] * f['H'] * f['W']
        return '%s %-6s %-20s M=%-8d cin=%-5d cout=%-5d taps=%d dil=%d' % (pn, kind, profiling.role(kind, f), M, f['cin'], f['cout'], f['taps'], f['dil'])
    dims = ' '.join('%s=%s' % (k, f[k]) for k in ('M', 'C', 'cols', 'rows', 'n', 'H', 'W', 'h', 'w', 'n_img') if isinstance(f.get(k), int) and f.get(k))
    return '%s %-14s %s' % (pn, kind, dims)
for pn, p in progs:
    for i, (kind, f) in enumerate(p.keepalive):
        key = sig(pn, kind, f)
        tag = table.setdefault(key, len(table) + 1)
        p.set_tag(i, tag)
        fl, by = profiling.work(kind, f)
        w = work.setdefault(tag, [0.0, 0.0, 0]); w[0] += fl; w[1] += by; w[2] += 1
for _ in range(a.steps): model.train_step(x, labels, refresh=False)
torch.cuda.synchronize()
nt = len(table) + 1; ms = [0.0] * nt; cnt = [0] * nt
for _, p in progs:
    m, c = p.read_timing(nt); ms = [u + v for u, v in zip(ms, m)]; cnt = [u + v for u, v in zip(cnt, c)]
rows = sorted(((ms[t] / a.steps, key, work[t]) for key, t in table.items()), reverse=True)
tot = sum(r[0] for r in rows)
print('total timed ms/step: %.2f over %d distinct ops' % (tot, len(rows)))
byk = {}
for msv, key, w in rows:
    k = ' '.join(key.split()[:2]); byk[k] = byk.get(k, 0.0) + msv
print('by program and kind:', ', '.join('%s %.2f' % kv for kv in sorted(byk.items(), key=lambda kv: -kv[1])))
print('%-100s %4s %9s %9s %8s %8s' % ('op', 'n', 'ms/step', 'avg us', 'TFLOP/s', 'GB/s'))
for msv, key, (fl, by, n) in rows[:a.top]:
    print('%-100s %4d %9.3f %9.1f %8.1f %8.1f' % (key, n, msv, 1e3 * msv / n, fl / (msv * 1e-3) / 1e12 if fl else 0, by / (msv * 1e-3) / 1e9 if by else 0))
