#!/usr/bin/env python
"""bf16 error of MobileNetDenseASPP against the fp32 oracle by image size, beside the oracle's own sensitivity to bf16
rounding of its conv weights and input (the yardstick of tests/test_mobilenet_cpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from e2e_common import make_batch  # noqa: E402
from test_mobilenet_cpu import bf16_yardstick, make_mobilenet_pair, rms_rel  # noqa: E402

torch.set_num_threads(16)
def calibrate(ref, x):
    """Running statistics := batch statistics of x (what a trained checkpoint's statistics are to its data)."""
    bns = [m for m in ref.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    keep = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    ref.train()
    with torch.no_grad():
        ref.forward_lowres(x)
    for m, k in zip(bns, keep):
        m.momentum = k
    ref.eval()


for randomize in ('calibrated', 'calibrated-affine', True, False):
    for H, W in ((64, 128), (256, 512), (1024, 2048)):
        ref, mod = make_mobilenet_pair()
        x, _ = make_batch(1, H, W)
        if randomize in ('calibrated', 'calibrated-affine'):
            if randomize == 'calibrated':
                for m in ref.modules():
                    if isinstance(m, torch.nn.BatchNorm2d):
                        m.reset_parameters()
            calibrate(ref, x)
            mod.load_state_dict(ref.state_dict())
        elif not randomize:
            for m in ref.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.reset_parameters()
                    m.reset_running_stats()
            mod.load_state_dict(ref.state_dict())
        x, _ = make_batch(1, H, W)
        ref.eval()
        mod = mod.cuda().eval()
        with torch.no_grad():
            yr = ref.forward_lowres(x)
            y16 = mod.forward_lowres(x.cuda()).cpu()
        yard = bf16_yardstick(ref, x)
        mod._engines.clear()
        mod.set_compute_dtype(torch.float32)
        with torch.no_grad():
            y32 = mod.forward_lowres(x.cuda()).cpu()
        print('randomized BN %s %4dx%4d: yardstick %.3e | bf16 vs oracle rms %.3e max %.3e | fp32 form vs oracle rms %.3e | logit rms %.3e'
              % (randomize, H, W, yard, rms_rel(yr, y16), float((yr - y16).abs().max() / yr.abs().max()), rms_rel(yr, y32),
                 float(yr.pow(2).mean().sqrt())), flush=True)
        del mod
        torch.cuda.empty_cache()
