#!/usr/bin/env python
"""Measures, on one MI355X, the numbers the full-size and trajectory parity tests assert (run before choosing tolerances):
  A  DenseASPP161 eval at 1x1024x2048 against the CPU oracle: bf16 and fp32 form
  B  DenseASPP161 one training step at 1x512x1024 against the CPU oracle
  C  loss trajectories (tiny cfg, DenseASPP121 at 128x256) over N Adam steps, and the bf16 gradient against the
     fp32 oracle's at the weights reached after those steps
Usage: python tools/explore_parity.py [A] [B] [C] > gpurun_out/explore_parity.txt
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from e2e_common import TINY_CFG, grad_report, make_batch, make_dropout_masks, make_pair, oracle_train  # noqa: E402


def say(*a):
    print(*a, flush=True)


def groups(g_ref, g_gpu, prefixes):
    out = {}
    for p in prefixes:
        ks = [k for k in g_ref if k.startswith(p)]
        if not ks:
            continue
        num = sum(float((g_ref[k] - g_gpu[k].float()).pow(2).sum()) for k in ks)
        den = sum(float(g_ref[k].pow(2).sum()) for k in ks)
        dot = sum(float((g_ref[k] * g_gpu[k].float()).sum()) for k in ks)
        n2 = sum(float(g_gpu[k].float().pow(2).sum()) for k in ks)
        out[p] = ((num / den) ** 0.5, dot / ((den * n2) ** 0.5 + 1e-30))
    return out


PREF = ('classification', 'ASPP_', 'features.norm5', 'features.transition4', 'features.denseblock4', 'features.transition3',
        'features.denseblock3', 'features.denseblock2', 'features.denseblock1', 'features.conv0')


def part_a():
    from cfgs import DenseASPP161
    torch.set_num_threads(16)
    ref, mod = make_pair(DenseASPP161.Model_CFG, randomize_bn=False)
    with torch.no_grad():
        ref.classification[1].weight.mul_(8.0)
    mod.load_state_dict(ref.state_dict())
    x, _ = make_batch(1, 1024, 2048, seed=3)
    ref.eval()
    t = time.time()
    with torch.no_grad():
        yr = ref.forward_lowres(x)
    say('A oracle forward 1x1024x2048: %.1f s' % (time.time() - t))
    mod = mod.cuda().eval()
    with torch.no_grad():
        y16 = mod.forward_lowres(x.cuda()).cpu()
    scale = float(yr.abs().max())
    e16 = float((yr - y16).abs().max()) / scale
    top2 = yr.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    clear = margin > 2 * 2e-2 * scale
    same = y16.argmax(1) == yr.argmax(1)
    say('A bf16 lowres logits err %.3e; argmax overall %.5f clear %.5f (clear frac %.3f)'
        % (e16, float(same.float().mean()), float(same[clear].float().mean()), float(clear.float().mean())))
    # full resolution too
    with torch.no_grad():
        yrf = F.interpolate(yr, scale_factor=8, mode='bilinear')
        yf = mod(x.cuda()).cpu()
    sf = float(yrf.abs().max())
    t2 = yrf.topk(2, dim=1).values
    cl = (t2[:, 0] - t2[:, 1]) > 2 * 2e-2 * sf
    sm = yf.argmax(1) == yrf.argmax(1)
    say('A bf16 full-res logits err %.3e; argmax overall %.5f clear %.5f (clear frac %.3f)'
        % (float((yrf - yf).abs().max()) / sf, float(sm.float().mean()), float(sm[cl].float().mean()), float(cl.float().mean())))
    mod._engines.clear()
    torch.cuda.empty_cache()
    mod.set_compute_dtype(torch.float32)
    t = time.time()
    with torch.no_grad():
        y32 = mod.forward_lowres(x.cuda()).cpu()
    torch.cuda.synchronize()
    say('A fp32 form err %.3e (%.1f s); argmax agreement all pixels %.6f'
        % (float((yr - y32).abs().max()) / scale, time.time() - t, float((y32.argmax(1) == yr.argmax(1)).float().mean())))


def part_b():
    from cfgs import DenseASPP161
    torch.set_num_threads(16)
    ref, mod = make_pair(DenseASPP161.Model_CFG, randomize_bn=False)
    B, H, W = 1, 512, 1024
    x, labels = make_batch(B, H, W, seed=4)
    masks = make_dropout_masks(ref, B, torch.Generator().manual_seed(5))
    t = time.time()
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    say('B oracle fwd+bwd 1x512x1024: %.1f s' % (time.time() - t))
    mod = mod.cuda().train()
    loss_g = mod.train_step(x.cuda(), labels.cuda(), dropout_masks={k: v.cuda() for k, v in masks.items()})
    torch.cuda.synchronize()
    g_gpu = {n: p.grad.cpu().clone() for n, p in mod.named_parameters()}
    say('B loss oracle %.6f gpu %.6f rel %.3e' % (float(loss_r), float(loss_g), abs(float(loss_g) - float(loss_r)) / abs(float(loss_r))))
    rel, cos, worst = grad_report(g_ref, g_gpu)
    say('B all params rel %.3e cos %.5f worst %s' % (rel, cos, worst))
    for k, v in groups(g_ref, g_gpu, PREF).items():
        say('B   %-28s rel %.3e cos %.5f' % (k, *v))
    br = dict(ref.named_buffers())
    worst_b = max((float((br[n] - b.cpu()).abs().max()), n) for n, b in mod.named_buffers() if b.dtype.is_floating_point)
    say('B running stats worst abs diff %.3e %s' % worst_b)


def trajectory(name, cfg, B, H, W, steps, lr, randomize_bn):
    from semantic_segmentation_amd.optim import FusedAdam
    ref, mod = make_pair(cfg, randomize_bn=randomize_bn)
    x, labels = make_batch(B, H, W, seed=7)
    mod = mod.cuda().train()
    ref.train()
    opt_r = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=1e-4)
    opt_g = FusedAdam(mod, lr=lr, weight_decay=1e-4)
    xg, lg = x.cuda(), labels.cuda()
    gen = torch.Generator().manual_seed(11)
    lr_l, lg_l = [], []
    for s in range(steps):
        masks = make_dropout_masks(ref, B, gen)
        opt_r.zero_grad()
        out = ref(x, masks)
        loss = F.cross_entropy(out, labels, ignore_index=255)
        loss.backward()
        opt_r.step()
        lr_l.append(float(loss))
        l2 = mod.train_step(xg, lg, dropout_masks={k: v.cuda() for k, v in masks.items()})
        opt_g.step()
        lg_l.append(float(l2))
    rel = [abs(a - b) / abs(a) for a, b in zip(lr_l, lg_l)]
    say('C %s lr %g: oracle loss %.4f -> %.4f, gpu %.4f -> %.4f; per-step rel diff max %.3e (step %d), final %.3e, mean %.3e'
        % (name, lr, lr_l[0], lr_l[-1], lg_l[0], lg_l[-1], max(rel), rel.index(max(rel)), rel[-1], sum(rel) / len(rel)))
    say('C   oracle:', ' '.join('%.4f' % v for v in lr_l))
    say('C   gpu   :', ' '.join('%.4f' % v for v in lg_l))
    # parameter drift between the two runs after the steps
    pr = dict(ref.named_parameters())
    num = sum(float((pr[n].detach() - p.detach().cpu()).pow(2).sum()) for n, p in mod.named_parameters())
    den = sum(float(p.detach().pow(2).sum()) for p in pr.values())
    say('C   parameter rel-L2 distance oracle vs gpu after %d steps: %.3e' % (steps, (num / den) ** 0.5))
    # gradient comparison at the weights the GPU run reached: oracle takes the GPU's state
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    ref.load_state_dict(sd)
    masks = make_dropout_masks(ref, B, gen)
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    loss_g = mod.train_step(xg, lg, dropout_masks={k: v.cuda() for k, v in masks.items()})
    torch.cuda.synchronize()
    g_gpu = {n: p.grad.cpu().clone() for n, p in mod.named_parameters()}
    r, c, worst = grad_report(g_ref, g_gpu)
    say('C   gradient at the reached weights: loss rel %.3e; all params rel %.3e cos %.5f worst %s'
        % (abs(float(loss_g) - float(loss_r)) / abs(float(loss_r)), r, c, worst))
    for k, v in groups(g_ref, g_gpu, PREF).items():
        say('C     %-28s rel %.3e cos %.5f' % (k, *v))


def part_c():
    from cfgs import DenseASPP121
    torch.set_num_threads(16)
    for lr in (1e-4, 1e-3):
        trajectory('tiny 2x64x96', TINY_CFG, 2, 64, 96, 40, lr, True)
        trajectory('121 1x128x256', DenseASPP121.Model_CFG, 1, 128, 256, 40, lr, False)
    trajectory('121 1x128x256 long', DenseASPP121.Model_CFG, 1, 128, 256, 150, 1e-3, False)


if __name__ == '__main__':
    parts = sys.argv[1:] or ['A', 'B', 'C']
    for p in parts:
        t0 = time.time()
        {'A': part_a, 'B': part_b, 'C': part_c}[p]()
        say('%s done in %.0f s' % (p, time.time() - t0))
