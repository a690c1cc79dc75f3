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

# stem: BN + ReLU + 3x3/2 max-pool of conv0's output (96 channels at 1/2 resolution) and its backward
H0, W0, C0 = 512, 1024, 96
M0, Mp = n * H0 * W0, n * (H0 // 2) * (W0 // 2)
x0 = torch.randn(M0 * C0, device=dev).to(bf); yp = torch.zeros(Mp * C0, device=dev).to(bf)
am = torch.zeros(Mp * C0, device=dev, dtype=torch.uint8); acc = torch.zeros(16 * 2 * C0, device=dev)
sc0, sh0, mu0, rs0 = vec(C0), vec(C0, -0.5), vec(C0, -0.5), vec(C0)
mp = dict(n_img=n, H=H0, W=W0, C=C0, x=Ref(x0), ldx=C0, scale=Ref(sc0), shift=Ref(sh0), y=Ref(yp), ldy=C0, argmax=Ref(am), stat_acc=Ref(acc),
          dy=None, lddy=0, dr=None, lddr=0, dr_f32=0, mean=None, rstd=None, s_acc=None, dtype=0)
timeit('maxpool_fwd 96 @1/2', 'maxpool_fwd', mp, M0 * C0 * 2 + Mp * C0 * 3)
dyp = torch.randn(Mp * C0, device=dev).to(bf); dr0 = torch.zeros(M0 * C0, device=dev).to(bf)
mb = dict(mp); mb.update(y=None, ldy=0, stat_acc=None, dy=Ref(dyp), lddy=C0, dr=Ref(dr0), lddr=C0, mean=Ref(mu0), rstd=Ref(rs0), s_acc=Ref(acc))
timeit('maxpool_bwd 96 @1/2', 'maxpool_bwd', mb, 2 * M0 * C0 * 2 + Mp * C0 * 3)
