"""-m gpu: the MobileNetDenseASPP kernels (3x3/2 stem, depthwise dilated 3x3, ReLU6 prologue, residual
bn_act) against the CPU kernel model through the C ABI, and the eval-mode network end to end."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import kernel_model  # noqa: E402
from e2e_common import make_batch  # noqa: E402
from optools import Ref, bf, close, run_both  # noqa: E402
from test_gpu_ops import check_conv, conv_fields  # noqa: E402
from test_mobilenet_cpu import bf16_yardstick, make_mobilenet_pair, rms_rel  # noqa: E402


@pytest.mark.parametrize('shape', [(2, 16, 24, 32), (1, 10, 6, 16), (3, 34, 18, 64)])
def test_stem3x3s2(shape):
    n, H, W, cout = shape
    gen = torch.Generator().manual_seed(3)
    ldy = cout + 8
    f = dict(n_img=n, H=H, W=W, x=Ref(torch.randn(n * 3 * H * W, generator=gen)),
             w=Ref(bf(torch.randn(cout * 27, generator=gen) * 27 ** -0.5)), cout=cout,
             y=Ref(torch.zeros(n * (H // 2) * (W // 2) * ldy, dtype=torch.bfloat16)), ldy=ldy)
    out = run_both('stem3', f)
    close(out['y'][0], out['y'][1], 1e-2, 'stem3.y')


DW_CASES = [
    # n, H, W, C, stride, dil, relu, affine
    (2, 12, 20, 32, 1, 1, 2, True),
    (1, 16, 24, 96, 2, 1, 2, True),
    (2, 9, 13, 144, 1, 2, 2, True),      # odd sizes; 18 channel groups (not a power of two)
    (1, 14, 10, 384, 1, 4, 2, True),
    (1, 20, 36, 576, 1, 8, 1, True),
    (1, 12, 40, 960, 1, 16, 2, True),    # dilation larger than the map height: most taps are padding
    (1, 7, 9, 24, 2, 1, 0, False),       # raw input, odd size with stride 2
]


@pytest.mark.parametrize('case', range(len(DW_CASES)))
def test_dwconv3x3(case):
    n, H, W, C, st, dil, relu, affine = DW_CASES[case]
    gen = torch.Generator().manual_seed(10 + case)
    ldx, ldy = C + 8, C
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    f = dict(n_img=n, H=H, W=W, C=C, stride=st, dil=dil, x=Ref(bf(torch.randn(n * H * W * ldx, generator=gen) * 2)), ldx=ldx,
             in_scale=None, in_shift=None, in_relu=relu, w=Ref(bf(torch.randn(C * 9, generator=gen) / 3)),
             y=Ref(torch.zeros(n * Ho * Wo * ldy, dtype=torch.bfloat16)), ldy=ldy)
    if affine:
        f['in_scale'] = Ref(torch.rand(C, generator=gen) + 0.5)
        f['in_shift'] = Ref(torch.randn(C, generator=gen) * 0.5)
    out = run_both('dwconv', f)
    close(out['y'][0], out['y'][1], 1e-2, 'dwconv.y')


@pytest.mark.parametrize('C,relu,res', [(16, 0, True), (24, 0, False), (320, 1, False), (160, 2, True)])
def test_bn_act_residual_relu6(C, relu, res):
    gen = torch.Generator().manual_seed(C)
    M, ldx, ldy, ldres = 333, C, C + 16, C + 8
    f = dict(M=M, C=C, x=Ref(bf(torch.randn(M * ldx, generator=gen) * 4)), ldx=ldx,
             scale=Ref(torch.rand(C, generator=gen) + 0.5), shift=Ref(torch.randn(C, generator=gen)), relu=relu,
             y=Ref(torch.zeros(M * ldy, dtype=torch.bfloat16)), ldy=ldy, stat_acc=None,
             res=Ref(bf(torch.randn(M * ldres, generator=gen))) if res else None, ldres=ldres if res else 0)
    out = run_both('bn_act', f)
    close(out['y'][0], out['y'][1], 1e-2, 'bn_act.y')


@pytest.mark.parametrize('cin,cout', [(32, 32), (16, 96), (96, 24), (144, 24), (960, 320), (24, 144)])
def test_pointwise_conv_relu6_prologue(cin, cout):
    f = conv_fields(2, 9, 13, cin, cout, 1, 1, seed=cin + cout)
    f['in_relu'] = 2
    f['in_scale'] = Ref(f['in_scale'].t * 3)      # push a good share of the inputs past the clamp at 6
    check_conv(f)


def test_mobilenet_eval_end_to_end():
    """GPU logits vs (a) the CPU kernel model with the same rounding points, (b) the fp32 oracle, the latter
    bounded by the oracle's own sensitivity to bf16 rounding of its weights (see bf16_yardstick)."""
    ref, mod = make_mobilenet_pair()
    _, mod_cpu = make_mobilenet_pair()
    mod_cpu.set_backend(kernel_model.ModelBackend())
    x, _ = make_batch(2, 64, 128)
    ref.eval(); mod_cpu.eval()
    mod = mod.cuda().eval()
    with torch.no_grad():
        yr = ref.forward_lowres(x)
        yc = mod_cpu.forward_lowres(x)
        yg = mod.forward_lowres(x.cuda()).cpu()
        full = mod(x.cuda()).cpu()
        am = mod.predict(x.cuda()).cpu()
    assert torch.isfinite(yg).all()
    yard = bf16_yardstick(ref, x)
    e_model, e_oracle = rms_rel(yc, yg), rms_rel(yr, yg)
    print('mobilenet e2e: gpu vs kernel model %.3e, gpu vs oracle %.3e, oracle bf16 yardstick %.3e' % (e_model, e_oracle, yard))
    assert e_model < 0.5 * yard + 1e-3, (e_model, yard)
    assert e_oracle < 2.0 * yard, (e_oracle, yard)
    # the full-resolution API and the fused arg-max agree with each other exactly where the top-1 margin is clear
    assert full.shape == (2, 19, 64, 128)
    top2 = full.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3 * float(full.abs().max())
    assert bool((am == full.argmax(1))[clear].all())
