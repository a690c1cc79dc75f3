#!/usr/bin/env python
"""Micro-benchmark of the row-wise streaming kernels at the shapes of the DenseASPP161 training step (diagnostic):
   python tools/rowwise_bench.py [--batch 32]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP  # noqa: F401
from semantic_segmentation_amd.backend import HipBackend
from semantic_segmentation_amd.engine import Ref
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--only', default=''); a = ap.parse_args()
dev = 'cuda'; be = HipBackend(); bf = torch.bfloat16
n = a.batch; M8 = n * 128 * 256


def timeit(name, kind, f, nbytes):
    if a.only and a.only not in name:
        return
    for _ in range(3): be.call(kind, **f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): be.call(kind, **f)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print('%-28s %9.1f us  %7.0f GB/s' % (name, us, nbytes / us / 1e3), flush=True)


def vec(c, lo=0.5): return torch.rand(c, device=dev) + lo

# BatchNorm-backward apply of a dense layer's norm2 (192 channels at 1/8 resolution)
C = 192
dr = torch.randn(M8 * C, device=dev).to(bf); y = torch.randn(M8 * C, device=dev).to(bf); dy = torch.zeros(M8 * C, device=dev).to(bf)
coef = torch.randn(3 * C, device=dev)
timeit('bn_bwd_apply 192 @1/8', 'bn_bwd_apply', dict(M=M8, C=C, dr=Ref(dr), lddr=C, dr_f32=0, y=Ref(y), ldy=C, coefA=Ref(coef, 0), coefB=Ref(coef, C),
       coefC=Ref(coef, 2 * C), dy=Ref(dy), lddy=C, colsum=None, dtype=0), 3 * M8 * C * 2)
