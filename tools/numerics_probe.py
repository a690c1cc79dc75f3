import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from e2e_common import make_pair, make_batch, make_dropout_masks, grad_report
from cfgs import DenseASPP121, DenseASPP161
import denseASPP
def run(cfg, H, W, B=1):
    ref, mod = make_pair(cfg, randomize_bn=False)
    x, labels = make_batch(B, H, W)
    masks = {k: v.cuda() for k, v in make_dropout_masks(ref, B, torch.Generator().manual_seed(5)).items()}
    sd = ref.state_dict()
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    l32 = float(mod.train_step(x.cuda(), labels.cuda(), dropout_masks=masks))
    g32 = {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    for slab, bn in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)):
        mod.load_state_dict(sd)
        mod.set_compute_dtype(torch.bfloat16)
        mod.grad_slab_dtype, mod.bn_grad_dtype = slab, bn
        mod._engines.clear()
        l16 = float(mod.train_step(x.cuda(), labels.cuda(), dropout_masks=masks))
        g16 = {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
        rel, cos, worst = grad_report(g32, g16)
        grp = {}
        for p in ('classification', 'ASPP_', 'features.denseblock4', 'features.denseblock3', 'features.denseblock1', 'features.conv0'):
            num = sum(float((g32[k].double() - g16[k].double()).pow(2).sum()) for k in g32 if k.startswith(p))
            den = sum(float(g32[k].double().pow(2).sum()) for k in g32 if k.startswith(p))
            grp[p] = '%.2e' % ((num / den) ** 0.5)
        print('%dx%d slab %s bn %s: rel %.3e cos %.5f loss %.5f/%.5f' % (H, W, str(slab)[6:], str(bn)[6:], rel, cos, l16, l32), grp, flush=True)
    del mod
    torch.cuda.empty_cache()
run(DenseASPP121.Model_CFG, 128, 256)
run(DenseASPP121.Model_CFG, 512, 1024)
