"""Golden vectors from the reference's OWN atrous conv + BatchNorm + ReLU blocks (tests/golden/bfp_adapt.pt).

The reference holds no DenseASPP source, but its BFP head contains the same operator pattern the DenseASPP path is
made of: 3x3 convolution with dilation 12 / padding 12, BatchNorm2d, ReLU (reference models/bfp_net.py:56-62,
BFPHead.adapt1 / adapt2).  This script imports the reference (read-only, CPU), instantiates BFPHead with a small
width, and records inputs, the modules' parameters and what the reference's modules compute:

    a   = adapt1[2](adapt1[1](y))         # BatchNorm (eval) + ReLU of a raw conv output y
    z   = adapt2[0](a)                    # 3x3, dilation 12, padding 12, no bias
    out = adapt2[2](adapt2[1](z))         # BatchNorm (eval) + ReLU

The fused kernel `BN scale/shift -> ReLU -> dilated 3x3 conv` must reproduce z from y, and bn_act must reproduce
out from z.  Run in the build container only (needs /root/reference):  python tests/golden/make_bfp_fixture.py
"""
import os
import sys

import torch
import torch.nn as nn

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
from models.bfp_net import BFPHead  # noqa: E402  (reference code)

torch.manual_seed(20)
head = BFPHead(256, 19, nn.BatchNorm2d).eval()            # inter_channels = 64
for bn in (head.adapt1[1], head.adapt2[1]):
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.3, 0.3)
    bn.running_mean.uniform_(-0.2, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
y = torch.randn(1, 64, 24, 32)
with torch.no_grad():
    a = head.adapt1[2](head.adapt1[1](y))
    z = head.adapt2[0](a)
    out = head.adapt2[2](head.adapt2[1](z))
conv = head.adapt2[0]
assert conv.dilation == (12, 12) and conv.padding == (12, 12) and conv.bias is None
fix = dict(y=y, z=z, out=out, conv_weight=conv.weight.data.clone(), dilation=12,
           bn1={k: v.clone() for k, v in head.adapt1[1].state_dict().items()}, bn1_eps=head.adapt1[1].eps,
           bn2={k: v.clone() for k, v in head.adapt2[1].state_dict().items()}, bn2_eps=head.adapt2[1].eps)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bfp_adapt.pt')
torch.save(fix, path)
print('wrote', path, os.path.getsize(path), 'bytes')
