#!/usr/bin/env python
"""Micro-benchmark of the MFMA kernels on the dominant DenseASPP161 layer shapes (diagnostic)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP
from semantic_segmentation_amd.backend import HipBackend
from semantic_segmentation_amd.engine import Ref
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=4); ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--only', default=''); ap.add_argument('--cin', type=int, default=1248)
ap.add_argument('--tile', default=''); ap.add_argument('--cout', type=int, default=192); a = ap.parse_args()
if a.tile: os.environ['DSASPP_CONV_TILE'] = a.tile
dev = 'cuda'; be = HipBackend(); R = 16
n, H, W = a.batch, 128, 256; M = n*H*W; LD = 2112; CIN = a.cin
bf = torch.bfloat16
slab = torch.randn(M*LD, device=dev).to(bf)
g = torch.zeros(M*LD, device=dev)
g16 = torch.zeros(M*LD, device=dev).to(bf)
y1 = torch.randn(M*192, device=dev).to(bf)
dz48 = torch.randn(M*48, device=dev).to(bf)
y512 = torch.randn(M*512, device=dev).to(bf); w512 = (torch.randn(2112*512, device=dev)*0.03).to(bf)
y1104 = torch.randn(M*1104, device=dev).to(bf); w1104 = (torch.randn(2112*1104, device=dev)*0.03).to(bf)
y1b, y1c, y1d = (torch.randn(M*192, device=dev).to(bf) for _ in range(3))
acc1, acc2, acc3 = (torch.zeros(R*2*LD, device=dev) for _ in range(3))
dr = torch.zeros(M*192, device=dev)
dr16 = torch.zeros(M*192, device=dev).to(bf)
def vec(c, lo=0.5): return torch.rand(c, device=dev) + lo
w1 = (torch.randn(192*CIN, device=dev)*0.03).to(bf); w1t = (torch.randn(192*CIN, device=dev)*0.03).to(bf)
w2 = (torch.randn(48*9*192, device=dev)*0.03).to(bf); w2t = (torch.randn(48*9*192, device=dev)*0.03).to(bf)
sc1, sh1, sc2, sh2 = vec(LD), vec(LD, -0.5), vec(192), vec(192, -0.5)
mean, rstd = vec(LD, -0.5), vec(LD)
acc = torch.zeros(R*2*LD, device=dev)
dw1 = torch.zeros(192*CIN, device=dev); dw2 = torch.zeros(48*9*192, device=dev)
base = dict(in_scale=None, in_shift=None, in_relu=0, in_mask=None, in_mask_ld=0, bias=None, out_mask=None, out_mask_ld=0, y=None, ldy=0,
            y_f32=0, stat_acc=None, epi_relu=0, ref=None, ldref=0, e_scale=None, e_shift=None, e_mean=None, e_rstd=None, s_acc=None, g=None, ldg=0, g_accumulate=0, g_bf16=0)
def conv(**kw):
    d = dict(base); d.update(kw); return ('conv', d)
cases = {
 'fwd1x1': conv(n_img=n, H=H, W=W, cin=CIN, cout=a.cout, taps=1, dil=1, x=Ref(slab), ldx=LD, w=Ref(w1), in_scale=Ref(sc1), in_shift=Ref(sh1), in_relu=1, y=Ref(y1), ldy=a.cout, stat_acc=Ref(acc)),
 'fwd3x3': conv(n_img=n, H=H, W=W, cin=192, cout=48, taps=9, dil=2, x=Ref(y1), ldx=192, w=Ref(w2), in_scale=Ref(sc2), in_shift=Ref(sh2), in_relu=1, y=Ref(slab, CIN), ldy=LD, stat_acc=Ref(acc)),
 'fwd3x3_d4': conv(n_img=n, H=H, W=W, cin=192, cout=48, taps=9, dil=4, x=Ref(y1), ldx=192, w=Ref(w2), in_scale=Ref(sc2), in_shift=Ref(sh2), in_relu=1, y=Ref(slab, CIN), ldy=LD, stat_acc=Ref(acc)),
 'dgrad3x3': conv(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=2, x=Ref(dz48), ldx=48, w=Ref(w2t), y=Ref(dr), ldy=192, y_f32=1, epi_relu=1, ref=Ref(y1), ldref=192, e_scale=Ref(sc2), e_shift=Ref(sh2), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc)),
 'dgrad3x3_noS': conv(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=2, x=Ref(dz48), ldx=48, w=Ref(w2t), y=Ref(dr), ldy=192, y_f32=1, epi_relu=1, ref=Ref(y1), ldref=192, e_scale=Ref(sc2), e_shift=Ref(sh2)),
 'dgrad3x3_plain': conv(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=2, x=Ref(dz48), ldx=48, w=Ref(w2t), y=Ref(dr), ldy=192, y_f32=1),
 'dgrad3x3_bf16': conv(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=2, x=Ref(dz48), ldx=48, w=Ref(w2t), y=Ref(y1b), ldy=192, y_f32=0, epi_relu=1, ref=Ref(y1), ldref=192, e_scale=Ref(sc2), e_shift=Ref(sh2), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc)),
 'dgrad1x1_slab': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g), ldg=LD, g_accumulate=1),
 'dgrad1x1_slab16': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g16), ldg=LD, g_accumulate=1, g_bf16=1),
 'slab_noS': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), g=Ref(g), ldg=LD, g_accumulate=1),
 'slab_noS_noref': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), g=Ref(g), ldg=LD, g_accumulate=1),
 'slab_store_only': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), g=Ref(g), ldg=LD, g_accumulate=0),
 'aspp_dgrad': conv(n_img=n, H=H, W=W, cin=512, cout=2112, taps=1, dil=1, x=Ref(y512), ldx=512, w=Ref(w512), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g16), ldg=LD, g_accumulate=1, g_bf16=1),
 'trans_dgrad': conv(n_img=n, H=H, W=W, cin=1104, cout=2112, taps=1, dil=1, x=Ref(y1104), ldx=1104, w=Ref(w1104), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g16), ldg=LD, g_accumulate=1, g_bf16=1),
 'slab_multi4': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g), ldg=LD, g_accumulate=1,
                     n_src=4, x1=Ref(y1b), w1=Ref(w1t), es1=Ref(sc1), eh1=Ref(sh1), sa1=Ref(acc1), x2=Ref(y1c), w2=Ref(w1t), es2=Ref(sc1), eh2=Ref(sh1), sa2=Ref(acc2), x3=Ref(y1d), w3=Ref(w1t), es3=Ref(sc1), eh3=Ref(sh1), sa3=Ref(acc3)),
 'slab_multi4_16': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g16), ldg=LD, g_accumulate=1, g_bf16=1,
                     n_src=4, x1=Ref(y1b), w1=Ref(w1t), es1=Ref(sc1), eh1=Ref(sh1), sa1=Ref(acc1), x2=Ref(y1c), w2=Ref(w1t), es2=Ref(sc1), eh2=Ref(sh1), sa2=Ref(acc2), x3=Ref(y1d), w3=Ref(w1t), es3=Ref(sc1), eh3=Ref(sh1), sa3=Ref(acc3)),
 'slab_multi4_noS': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), g=Ref(g), ldg=LD, g_accumulate=1,
                     n_src=4, x1=Ref(y1b), w1=Ref(w1t), es1=Ref(sc1), eh1=Ref(sh1), x2=Ref(y1c), w2=Ref(w1t), es2=Ref(sc1), eh2=Ref(sh1), x3=Ref(y1d), w3=Ref(w1t), es3=Ref(sc1), eh3=Ref(sh1)),
 'slab_multi2': conv(n_img=n, H=H, W=W, cin=192, cout=CIN, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t), epi_relu=1, ref=Ref(slab), ldref=LD, e_scale=Ref(sc1), e_shift=Ref(sh1), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), g=Ref(g), ldg=LD, g_accumulate=1,
                     n_src=2, x1=Ref(y1b), w1=Ref(w1t), es1=Ref(sc1), eh1=Ref(sh1), sa1=Ref(acc1)),
 'wgrad1x1': ('wgrad', dict(n_img=n, H=H, W=W, cin=CIN, cout=192, taps=1, dil=1, x=Ref(slab), ldx=LD, in_scale=Ref(sc1), in_shift=Ref(sh1), in_relu=1, in_mask=None, in_mask_ld=0, dz=Ref(y1), lddz=192, dw=Ref(dw1), ksplit=0)),
 'trans_wgrad': ('wgrad', dict(n_img=n, H=H, W=W, cin=CIN, cout=1104, taps=1, dil=1, x=Ref(slab), ldx=LD, in_scale=Ref(sc1), in_shift=Ref(sh1), in_relu=1, in_mask=None, in_mask_ld=0, dz=Ref(y1104), lddz=1104, dw=Ref(torch.zeros(1104*CIN, device=dev)), ksplit=0)),
 'aspp_wgrad': ('wgrad', dict(n_img=n, H=H, W=W, cin=CIN, cout=512, taps=1, dil=1, x=Ref(slab), ldx=LD, in_scale=Ref(sc1), in_shift=Ref(sh1), in_relu=1, in_mask=None, in_mask_ld=0, dz=Ref(y512), lddz=512, dw=Ref(torch.zeros(512*CIN, device=dev)), ksplit=0)),
 'wgrad3x3': ('wgrad', dict(n_img=n, H=H, W=W, cin=192, cout=48, taps=9, dil=2, x=Ref(y1), ldx=192, in_scale=Ref(sc2), in_shift=Ref(sh2), in_relu=1, in_mask=None, in_mask_ld=0, dz=Ref(dz48), lddz=48, dw=Ref(dw2), ksplit=0)),
}
for cn in (48, 96, 144):      # in-group data gradient: a narrow column range at channel CIN - cn of the slab
    c_lo = CIN - cn
    cases['narrow%d' % cn] = conv(n_img=n, H=H, W=W, cin=192, cout=cn, taps=1, dil=1, x=Ref(y1), ldx=192, w=Ref(w1t, c_lo * 192), epi_relu=1, ref=Ref(slab, c_lo), ldref=LD,
                                  e_scale=Ref(sc1, c_lo), e_shift=Ref(sh1, c_lo), e_mean=Ref(mean, c_lo), e_rstd=Ref(rstd, c_lo), s_acc=Ref(acc, c_lo), s_ld=CIN, g=Ref(g16, c_lo), ldg=LD,
                                  g_accumulate=1, g_bf16=1)
corr = torch.randn(2*LD, device=dev) * 0.1
for dl in (1, 2, 4):          # 3x3 data gradient with the slab read-out as its prologue
    c_lo = CIN - 48
    cases['dgrad3x3_fused_d%d' % dl] = conv(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=dl, x=Ref(g16, c_lo), ldx=LD, w=Ref(w2t), y=Ref(dr16), ldy=192, epi_relu=1, ref=Ref(y1), ldref=192,
        e_scale=Ref(sc2), e_shift=Ref(sh2), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), gx_corr1=Ref(corr, c_lo), gx_corr2=Ref(corr, LD + c_lo), gx_xs=Ref(slab, c_lo), gx_ldxs=LD,
        gx_mean=Ref(mean, c_lo), gx_rstd=Ref(rstd, c_lo), gx_dz=Ref(dz48), gx_lddz=48)
w3a = (torch.randn(128*9*512, device=dev)*0.02).to(bf); sc512, sh512 = vec(512), vec(512, -0.5); b128 = vec(128, -0.5)
for dl in (3, 12, 24):        # ASPP 3x3 forward: 512 -> 128, written into the ASPP slab
    cases['aspp_fwd3x3_d%d' % dl] = conv(n_img=n, H=H, W=W, cin=512, cout=128, taps=9, dil=dl, x=Ref(y512), ldx=512, w=Ref(w3a), in_scale=Ref(sc512), in_shift=Ref(sh512),
                                         in_relu=1, bias=Ref(b128), y=Ref(slab, 1024), ldy=LD, stat_acc=Ref(acc))
dwg = [torch.zeros(192*CIN, device=dev) for _ in range(4)]
def wgroup(nsrc, ks=0):
    f = dict(n_img=n, H=H, W=W, cin=CIN, cout=192, taps=1, dil=1, x=Ref(slab), ldx=LD, in_scale=Ref(sc1), in_shift=Ref(sh1), in_relu=1, in_mask=None,
             in_mask_ld=0, dz=Ref(y1), lddz=192, dw=Ref(dwg[0]), ksplit=ks, n_src=nsrc)
    for k, yk in list(enumerate((y1, y1b, y1c, y1d)))[1:nsrc]:
        f.update({'dz%d' % k: Ref(yk), 'dw%d' % k: Ref(dwg[k]), 'sc%d' % k: Ref(sc1), 'sh%d' % k: Ref(sh1), 'cin%d' % k: CIN - 48 * k})
    return ('wgrad', f)
for ns in (2, 3, 4):
    cases['wgroup%d' % ns] = wgroup(ns)
for dbg in (2, 4, 8, 6):
    kind, f = cases['fwd3x3']; f = dict(f); f['in_mask_ld'] = dbg; cases['fwd3x3_dbg%d' % dbg] = (kind, f)
for ks in (25, 38, 51, 76):
    kind, f = cases['wgrad1x1']; f = dict(f); f['ksplit'] = ks; cases['wgrad1x1_ks%d' % ks] = (kind, f)
for ks in ():
    kind, f = cases['wgrad3x3']; f = dict(f); f['ksplit'] = ks; cases['wgrad3x3_ks%d' % ks] = (kind, f)
for name, (kind, f) in cases.items():
    if a.only and not any(name == o or (o.endswith('*') and name.startswith(o[:-1])) for o in a.only.split(',')): continue
    from semantic_segmentation_amd.engine import OpList
    ops = OpList()
    for _ in range(a.iters): ops.op(kind, **f)
    prog = be.compile(ops)            # replayed by the C++ runner: no per-launch Python time in the measurement
    prog.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); prog.run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/a.iters
    fl = 2.0*M*f['cout']*f['taps']*(sum([f['cin']] + [f['cin%d' % k] for k in range(1, f['n_src'])]) if kind == 'wgrad' and f.get('n_src', 0) > 1 else f['cin']*max(1, f.get('n_src', 0)))
    print('%-14s %9.1f us  %7.1f TFLOP/s' % (name, us, fl/us/1e6), flush=True)

# ---- do two different kernels overlap when they run on two streams?  (python tools/conv_bench.py --only overlap)
if a.only == 'overlap':
    from semantic_segmentation_amd.engine import OpList
    kind, f = cases['fwd1x1']; f = dict(f); f['stat_acc'] = None
    pairs = {'fwd1x1': (kind, f), 'wgrad1x1': cases['wgrad1x1'], 'dgrad3x3': cases['dgrad3x3'], 'wgrad3x3': cases['wgrad3x3'],
             'slab_multi4': cases['slab_multi4']}
    progs = {}
    for name, (kind, f) in pairs.items():
        ops = OpList()
        for _ in range(a.iters): ops.op(kind, **f)
        progs[name] = be.compile(ops)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def wall(fn):
        fn(); torch.cuda.synchronize()
        t = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t) * 1e6 / a.iters
    for x, y in (('fwd1x1', 'wgrad1x1'), ('dgrad3x3', 'wgrad3x3'), ('slab_multi4', 'wgrad1x1')):
        tx = wall(lambda: progs[x].run(stream=s1.cuda_stream)); ty = wall(lambda: progs[y].run(stream=s1.cuda_stream))
        def both():
            progs[x].run(stream=s1.cuda_stream); progs[y].run(stream=s2.cuda_stream)
        tb = wall(both)
        print('%-12s %7.1f us + %-10s %7.1f us = %7.1f sequential; two streams: %7.1f us (%.2fx)' % (x, tx, y, ty, tx + ty, tb, (tx + ty) / tb), flush=True)
