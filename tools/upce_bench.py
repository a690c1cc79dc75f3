#!/usr/bin/env python
"""Micro-benchmark of the fused bilinear x8 + cross-entropy kernel at the benchmark size (diagnostic)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP  # noqa: F401
from semantic_segmentation_amd.backend import HipBackend
from semantic_segmentation_amd.engine import OpList, Ref
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16); ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
be = HipBackend(); dev = 'cuda'
n, h, w, nc, ldl = a.batch, 128, 256, 19, 32
M = n * h * w
logits = torch.randn(M * ldl, device=dev) * 2
labels = torch.randint(0, nc, (n * 64 * h * w,), device=dev)
labels[torch.rand(labels.shape, device=dev) < 0.1] = 255
f = dict(n_img=n, h=h, w=w, n_class=nc, ldl=ldl, logits=Ref(logits), labels=Ref(labels), ignore_index=255,
         dlogits=Ref(torch.zeros(M * ldl, device=dev)), accum=Ref(torch.zeros(1024, device=dev)), grad_scale=1.0, loss=Ref(torch.zeros(1, device=dev)),
         dlogits_bf16=Ref(torch.zeros(M * ldl, dtype=torch.bfloat16, device=dev)), lddb=ldl, dbias=Ref(torch.zeros(32, device=dev)))
ops = OpList()
for _ in range(a.iters): ops.op('upce', **f)
prog = be.compile(ops); prog.run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); prog.run(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
by = M * nc * 4.0 + M * 64 * 8.0 + M * nc * 4.0
print('upce batch %d: %.1f us, %.1f GB/s algorithmic' % (n, us, by / us / 1e3))
