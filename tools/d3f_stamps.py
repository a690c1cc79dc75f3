#!/usr/bin/env python
"""Phase profile of the fused 3x3 data gradient from a -DD3F_STAMP=1 build (DSASPP_LIB_PATH): shader-clock stamps of wave 0 of every block."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import denseASPP  # noqa: F401
from semantic_segmentation_amd import _lib
from semantic_segmentation_amd.backend import HipBackend
from semantic_segmentation_amd.engine import OpList, Ref
be = HipBackend(); dev = 'cuda'; bf = torch.bfloat16
n, H, W, LD, dl = 16, 128, 256, 2112, int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = n * H * W; c_lo = 1200
g16 = torch.randn(M * LD, device=dev).to(bf); slab = torch.randn(M * LD, device=dev).to(bf)
y1 = torch.randn(M * 192, device=dev).to(bf); dr16 = torch.zeros(M * 192, device=dev).to(bf); dz48 = torch.zeros(M * 48, device=dev).to(bf)
w2t = (torch.randn(48 * 9 * 192, device=dev) * 0.03).to(bf)
vec = lambda c, lo=0.5: torch.rand(c, device=dev) + lo
sc2, sh2, mean, rstd, corr, acc = vec(192), vec(192, -0.5), vec(LD, -0.5), vec(LD), torch.randn(2 * LD, device=dev) * 0.1, torch.zeros(16 * 2 * LD, device=dev)
f = dict(n_img=n, H=H, W=W, cin=48, cout=192, taps=9, dil=dl, x=Ref(g16, c_lo), ldx=LD, w=Ref(w2t), y=Ref(dr16), ldy=192, epi_relu=1, ref=Ref(y1), ldref=192,
         e_scale=Ref(sc2), e_shift=Ref(sh2), e_mean=Ref(mean), e_rstd=Ref(rstd), s_acc=Ref(acc), gx_corr1=Ref(corr, c_lo), gx_corr2=Ref(corr, LD + c_lo),
         gx_xs=Ref(slab, c_lo), gx_ldxs=LD, gx_mean=Ref(mean, c_lo), gx_rstd=Ref(rstd, c_lo), gx_dz=Ref(dz48), gx_lddz=48)
for _ in range(3): be.call('conv', **f)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.dsaspp_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.dsaspp_debug_read_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(4096, 16)[:2048].astype(np.int64)
names = ['tables+W0 issue', 'stage patch', 'to phase0 barrier', 'store W0+barrier', 'phase0 MFMA', 'barrier', 'store W1+barrier', 'phase1 MFMA', 'barrier', 'store W2+barrier',
         'phase2 MFMA', 'ref loads+barrier', 'epilogue', 'barrier']
d = np.diff(st[:, :15], axis=1)
print('dilation %d: median cycles of wave 0 per phase over %d blocks (shader clock)' % (dl, st.shape[0]))
for i, nm in enumerate(names):
    print('  %-22s %8.0f  (p10 %8.0f, p90 %8.0f)' % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print('  %-22s %8.0f' % ('block total', np.median(st[:, 14] - st[:, 0])))
t0 = st[:, 0].min(); print('  kernel span %.0f cycles; blocks start p50 %.0f, last start %.0f' % (st[:, 14].max() - t0, np.median(st[:, 0] - t0), (st[:, 0] - t0).max()))
