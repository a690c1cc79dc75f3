"""Golden vectors from more of the reference's OWN modules (tests/golden/reference_ops.pt).

  * Bottleneck of the dilated ResNet (reference models/dilated_model/resnet.py:61-110) with dilation 2 and 4, the
    cousin of a dense layer: conv1x1 -> BatchNorm -> ReLU -> dilated conv3x3 -> BatchNorm -> ReLU.  Recorded in
    train() mode (batch statistics, running-statistic update, and the gradients autograd gives for a fixed upstream
    gradient) and in eval() mode (running statistics).
  * _fast_hist / evaluate (reference miscellaneous/misc.py:62-90): the confusion matrix and the scores the
    reference derives from it, on random predictions and labels with ignore pixels.
  * nn.Upsample(scale_factor=8, mode='bilinear') + CrossEntropyLoss(ignore_index=255), the operators of the
    reference's loss line (graph_seg_main.py:216,238): loss and gradient w.r.t. the low-resolution scores.

Run in the build container only (needs /root/reference):  python tests/golden/make_reference_ops_fixture.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
from models.dilated_model.resnet import Bottleneck  # noqa: E402  (reference code)
from miscellaneous.misc import _fast_hist, evaluate  # noqa: E402  (reference code)

fix = {}
for dil in (2, 4):
    torch.manual_seed(40 + dil)
    blk = Bottleneck(64, 16, stride=1, dilation=dil, downsample=None, previous_dilation=dil, norm_layer=nn.BatchNorm2d)
    for bn in (blk.bn1, blk.bn2):
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 64, 12, 16)
    up = torch.randn(2, 16, 12, 16)                       # upstream gradient w.r.t. relu(bn2(conv2(.)))
    rec = dict(x=x, up=up, dilation=dil, conv1_weight=blk.conv1.weight.data.clone(), conv2_weight=blk.conv2.weight.data.clone(),
               eps=blk.bn1.eps, momentum=blk.bn1.momentum,
               bn1_before={k: v.clone() for k, v in blk.bn1.state_dict().items()},
               bn2_before={k: v.clone() for k, v in blk.bn2.state_dict().items()})
    # eval mode first (does not touch the running statistics)
    blk.eval()
    with torch.no_grad():
        y1 = blk.conv1(x)
        a1 = blk.relu(blk.bn1(y1))
        z = blk.conv2(a1)
        a2 = blk.relu(blk.bn2(z))
    rec['eval'] = dict(y1=y1, z=z, a2=a2.clone())
    # train mode: the first half of Bottleneck.forward, statement by statement, with gradients
    blk.train()
    blk.zero_grad()
    xg = x.clone().requires_grad_(True)
    y1 = blk.conv1(xg)
    n1 = blk.bn1(y1)
    a1 = F.relu(n1)
    z = blk.conv2(a1)
    n2 = blk.bn2(z)
    a2 = F.relu(n2)
    for t in (y1, z, a1):
        t.retain_grad()
    (a2 * up).sum().backward()
    rec['train'] = dict(y1=y1.detach().clone(), z=z.detach().clone(), a2=a2.detach().clone(), dz=z.grad.clone(), da1=a1.grad.clone(),
                        dy1=y1.grad.clone(), dx=xg.grad.clone(), dconv1=blk.conv1.weight.grad.clone(),
                        dconv2=blk.conv2.weight.grad.clone(), dbn1_w=blk.bn1.weight.grad.clone(), dbn1_b=blk.bn1.bias.grad.clone(),
                        dbn2_w=blk.bn2.weight.grad.clone(), dbn2_b=blk.bn2.bias.grad.clone(),
                        bn1_after={k: v.clone() for k, v in blk.bn1.state_dict().items()},
                        bn2_after={k: v.clone() for k, v in blk.bn2.state_dict().items()})
    fix['bottleneck_d%d' % dil] = rec

rng = np.random.RandomState(7)
pred = rng.randint(0, 19, size=(2, 64, 96))
true = rng.randint(0, 19, size=(2, 64, 96))
true[rng.rand(2, 64, 96) < 0.12] = 255
hist = sum(_fast_hist(p.flatten(), t.flatten(), 19) for p, t in zip(pred, true))
acc, acc_cls, acc_cls_mean, mean_iu, fwavacc = evaluate(list(pred), list(true), 19)
fix['fast_hist'] = dict(pred=torch.from_numpy(pred.astype(np.uint8)), true=torch.from_numpy(true.astype(np.uint8)), hist=torch.from_numpy(hist.astype(np.int64)),
                        acc=float(acc), acc_cls_mean=float(acc_cls_mean), mean_iu=float(mean_iu), fwavacc=float(fwavacc))

torch.manual_seed(3)
logits = (torch.randn(1, 19, 8, 16) * 2).requires_grad_(True)
labels = torch.randint(0, 19, (1, 64, 128))
labels[torch.rand(1, 64, 128) < 0.1] = 255
full = nn.Upsample(scale_factor=8, mode='bilinear')(logits)
loss = nn.CrossEntropyLoss(ignore_index=255)(full, labels)
loss.backward()
fix['upsample_ce'] = dict(logits=logits.detach().clone(), labels=labels.to(torch.uint8), full_row0=full.detach()[0, :, 0, :].clone(), loss=loss.detach().clone(),
                          dlogits=logits.grad.clone())

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_ops.pt')
torch.save(fix, path)
print('wrote', path, os.path.getsize(path), 'bytes')
