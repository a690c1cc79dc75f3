"""Shared set-up for end-to-end comparisons: oracle (fp32 torch.nn) vs the product module."""
import torch
import torch.nn.functional as F

import denseASPP
from conftest import TINY_CFG  # noqa: F401
from oracle.dense_aspp_ref import DenseASPPRef, make_dropout_masks


def make_pair(cfg, seed=0, randomize_bn=True, output_stride=8):
    torch.manual_seed(seed)
    ref = DenseASPPRef(cfg, output_stride=output_stride)
    if randomize_bn:
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    mod = denseASPP.DenseASPP(cfg, 19, output_stride)
    mod.load_state_dict(ref.state_dict())
    return ref, mod


def make_batch(B, H, W, seed=1, n_class=19):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g).clamp(-2.2, 2.7)
    labels = torch.randint(0, n_class, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return x, labels


def oracle_train(ref, x, labels, masks):
    ref.train()
    ref.zero_grad()
    out = ref(x, masks)
    loss = F.cross_entropy(out, labels, ignore_index=255)
    loss.backward()
    return loss.detach(), {n: p.grad.clone() for n, p in ref.named_parameters()}


def grad_report(g_ref, g_test):
    """Global relative L2 error and cosine similarity over all parameters, plus the per-tensor worst."""
    num = den = dot = n2 = 0.0
    worst = (0.0, '')
    for n, gr in g_ref.items():
        gt = g_test[n].float()
        num += float((gr - gt).pow(2).sum())
        den += float(gr.pow(2).sum())
        dot += float((gr * gt).sum())
        n2 += float(gt.pow(2).sum())
        rel = float((gr - gt).norm() / (gr.norm() + 1e-12))
        if gr.norm() > 1e-6 and rel > worst[0]:
            worst = (rel, n)
    return (num / den) ** 0.5, dot / ((den * n2) ** 0.5 + 1e-30), worst


__all__ = ['make_pair', 'make_batch', 'oracle_train', 'grad_report', 'make_dropout_masks', 'TINY_CFG']
