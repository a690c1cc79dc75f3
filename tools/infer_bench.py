#!/usr/bin/env python
"""Batch-sharded inference throughput (no collective) of any cfg on one MI355X: forward + fused x8 upsample/arg-max.

    python tools/infer_bench.py --model MobileNetDenseASPP --batch 16                      # BASELINE config 2
    python tools/infer_bench.py --model DenseASPP169 --batch 4 --height 2048 --width 4096  # config 5 (halo stress)

Prints one JSON line: images/s, per-op-class HIP-event times with the algorithmic flops / bytes of each class.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='MobileNetDenseASPP')
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--height', type=int, default=1024)
ap.add_argument('--width', type=int, default=2048)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--cpu-baseline', action='store_true', help='time the fp32 torch.nn oracle on the host cores (batch 1)')
a = ap.parse_args()

import denseASPP  # noqa: E402
from semantic_segmentation_amd import profiling  # noqa: E402

cfg = importlib.import_module('cfgs.' + a.model).Model_CFG
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = denseASPP.DenseASPP(cfg).to(dev).eval()
B, H, W = a.batch, a.height, a.width
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, H, W, generator=g).clamp_(-2.2, 2.7).to(dev)
eng = model.engine(B, H, W, False)
model.refresh_weights()
amx = eng.prog_argmax()


def step():
    eng.set_input(x)
    eng._fwd.run()
    amx.run()


def op_key(kind, f):
    if kind == 'conv':
        t = profiling.kernel_class(kind, f)
        return profiling.CLASSES[t] + (' 3x3' if f['taps'] == 9 else ' 1x1')
    return kind


def op_work(kind, f):
    """(flops, algorithmic HBM bytes) of one launch."""
    if kind == 'conv':
        M = f['n_img'] * f['H'] * f['W']
        return 2.0 * M * f['cout'] * f['taps'] * f['cin'], 2.0 * M * (f['cin'] + f['cout'])
    if kind == 'dwconv':
        mi = f['n_img'] * f['H'] * f['W']
        mo = f['n_img'] * ((f['H'] - 1) // f['stride'] + 1) * ((f['W'] - 1) // f['stride'] + 1)
        return 18.0 * mo * f['C'], 2.0 * f['C'] * (mi + mo)
    if kind == 'stem3':
        mo = f['n_img'] * (f['H'] // 2) * (f['W'] // 2)
        return 54.0 * mo * f['cout'], 12.0 * f['n_img'] * f['H'] * f['W'] + 2.0 * mo * f['cout']
    if kind == 'bn_act':
        return 0.0, 2.0 * f['M'] * f['C'] * (3 if f.get('res') is not None else 2)
    if kind == 'up_argmax':
        m = f['n_img'] * f['h'] * f['w']
        return 0.0, m * f['n_class'] * 4.0 + m * 64 * 8.0
    return 0.0, 0.0


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
table, per = {}, {}
for prog in (eng._fwd, amx):
    for i, (kind, f) in enumerate(prog.keepalive):
        if kind in ('bn_finalize', 'memset'):
            continue
        tag = table.setdefault(op_key(kind, f), len(table) + 1)
        prog.set_tag(i, tag)
        fl, by = op_work(kind, f)
        p = per.setdefault(tag, [0.0, 0.0, 0])
        p[0] += fl; p[1] += by; p[2] += 1
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = time.time() - t0
nt = len(table) + 1
ms, cnt = [0.0] * nt, [0] * nt
for prog in (eng._fwd, amx):
    m, c = prog.read_timing(nt)
    ms = [p + q for p, q in zip(ms, m)]
    cnt = [p + q for p, q in zip(cnt, c)]
kernels = {}
for key, tag in table.items():
    fl, by, n = per[tag]
    sec = ms[tag] * 1e-3 / a.steps
    kernels[key] = dict(launches=n, ms_per_step=round(ms[tag] / a.steps, 3),
                        tflops=round(fl / sec / 1e12, 1) if fl and sec else None,
                        alg_gbs=round(by / sec / 1e9, 1) if by and sec else None)
macs = model.conv_macs(H, W)
value = B * a.steps / dt
out = dict(metric='%s inference img/s @%dx%d' % (a.model, H, W), value=round(value, 2), unit='img/s', n_gpus=1,
           steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * dt / a.steps, 3), batch=B, dtype='bf16', data='synthetic',
           gmacs_per_image=round(macs / 1e9, 2), model_tflops_per_s=round(2.0 * macs * value / 1e12, 1),
           hbm_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           kernels=dict(sorted(kernels.items(), key=lambda kv: -kv[1]['ms_per_step'])))
if a.cpu_baseline:
    if model.spec.kind == 'mobilenet':
        from oracle.mobilenet_dense_aspp_ref import MobileNetDenseASPPRef as RefNet
    else:
        from oracle.dense_aspp_ref import DenseASPPRef as RefNet
    threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    ref = RefNet(cfg).eval()
    xc = x[:1].cpu()
    with torch.no_grad():
        ref(xc[:, :, :128, :256])
        t = time.time()
        ref(xc)
        t = time.time() - t
    out['cpu_baseline'] = dict(value=round(1.0 / t, 4), unit='img/s', cores=threads, kind='port',
                               sample='1 eval forward, batch 1, %dx%d, fp32 torch.nn oracle (%.1f s)' % (H, W, t))
print(json.dumps(out))
