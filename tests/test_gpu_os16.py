"""-m gpu: output_stride = 16 (x2 feature upsample kernels and the network end to end)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_common import TINY_CFG, make_batch, make_dropout_masks, make_pair, oracle_train  # noqa: E402
from optools import Ref, bf, close, run_both  # noqa: E402

R = 16


@pytest.mark.parametrize('shape', [(2, 5, 7, 48), (1, 8, 16, 200), (3, 1, 4, 16)])
def test_upsample2_fwd_bwd(shape):
    n, h, w, C = shape
    gen = torch.Generator().manual_seed(C)
    ldx, ldy = C + 8, C + 16
    f = dict(n_img=n, h=h, w=w, C=C, x=Ref(bf(torch.randn(n * h * w * ldx, generator=gen))), ldx=ldx,
             scale=Ref(torch.rand(C, generator=gen) + 0.5), shift=Ref(torch.randn(C, generator=gen) * 0.3), relu=1,
             y=Ref(torch.zeros(n * 4 * h * w * ldy, dtype=torch.bfloat16)), ldy=ldy, stat_acc=Ref(torch.zeros(R * 2 * C)),
             dx=None, lddx=0, dx_f32=0, mean=None, rstd=None, s_acc=None, dtype=0)
    out = run_both('up2_fwd', f)
    close(out['y'][0], out['y'][1], 1e-2, 'up2_fwd.y')
    close(out['stat_acc'][0].view(R, -1).sum(0), out['stat_acc'][1].view(R, -1).sum(0), 2e-2, 'up2_fwd.stat_acc')
    for dx_f32 in (0, 1):
        b = dict(n_img=n, h=h, w=w, C=C, x=Ref(bf(torch.randn(n * h * w * ldx, generator=gen))), ldx=ldx, scale=None,
                 shift=None, relu=0, y=Ref(bf(torch.randn(n * 4 * h * w * ldy, generator=gen))), ldy=ldy, stat_acc=None,
                 dx=Ref(torch.zeros(n * h * w * C, dtype=torch.float32 if dx_f32 else torch.bfloat16)), lddx=C, dx_f32=dx_f32,
                 mean=Ref(torch.randn(C, generator=gen) * 0.1), rstd=Ref(torch.rand(C, generator=gen) + 0.5),
                 s_acc=Ref(torch.zeros(R * 2 * C)), dtype=0)
        out = run_both('up2_bwd', b)
        close(out['dx'][0], out['dx'][1], 1e-2, 'up2_bwd.dx')
        close(out['s_acc'][0].view(R, -1).sum(0), out['s_acc'][1].view(R, -1).sum(0), 2e-2, 'up2_bwd.s_acc')


def test_os16_eval_and_train_match_oracle():
    ref, mod = make_pair(TINY_CFG, output_stride=16)
    x, labels = make_batch(2, 64, 96)
    ref.eval()
    mod = mod.cuda().eval()
    with torch.no_grad():
        yr = ref(x)
        ym = mod(x.cuda()).cpu()
    assert float((yr - ym).abs().max() / yr.abs().max()) < 2e-2
    mod.set_compute_dtype(torch.float32)
    with torch.no_grad():
        y32 = mod(x.cuda()).cpu()
    assert float((yr - y32).abs().max() / yr.abs().max()) < 1e-4
    mod.set_compute_dtype(torch.bfloat16)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod.train()
    loss = mod.train_step(x.cuda(), labels.cuda(), dropout_masks={k: v.cuda() for k, v in masks.items()})
    assert abs(float(loss) - float(loss_r)) / float(loss_r) < 5e-3
    # gradients: tests/test_gpu_e2e.py::test_train_step_tiny[16] (HIP backward vs the CPU kernel model from an identical
    # forward state, and the oracle bounded by its own bf16 sensitivity)
