"""-m gpu: the fp32 form of the eval-mode forward pass (descriptor dtype = 1, model.set_compute_dtype(torch.float32)).

Bar: logits within 1e-4 (max abs error / max |reference|) of the fp32 oracle on the same inputs and weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import kernel_model  # noqa: E402
from e2e_common import TINY_CFG, make_batch, make_pair  # noqa: E402
from optools import Ref, close, run_both  # noqa: E402
from test_mobilenet_cpu import make_mobilenet_pair  # noqa: E402

FP32_TOL = 1e-4


@pytest.fixture
def fp32_model():
    kernel_model.set_act_dtype(torch.float32)
    yield
    kernel_model.set_act_dtype(torch.bfloat16)


CASES = [
    # n, H, W, cin, cout, taps, dil, relu, bias
    (2, 9, 13, 144, 192, 1, 1, 1, False),
    (1, 16, 24, 72, 48, 9, 2, 1, True),        # cin not a multiple of the 16-wide K-step
    (2, 12, 20, 64, 128, 9, 6, 1, True),
    (1, 8, 16, 176, 32, 1, 1, 0, True),
    (2, 10, 6, 24, 144, 1, 1, 2, False),       # ReLU6 prologue, narrow input
    (1, 7, 5, 320, 19 + 13, 1, 1, 0, True),    # classifier-like: 32 padded rows
]


@pytest.mark.parametrize('case', range(len(CASES)))
def test_conv_fp32(case, fp32_model):
    n, H, W, cin, cout, taps, dil, relu, bias = CASES[case]
    gen = torch.Generator().manual_seed(40 + case)
    M, ldx, ldy = n * H * W, cin + 8, cout + 4
    f = dict(n_img=n, H=H, W=W, cin=cin, cout=cout, taps=taps, dil=dil, x=Ref(torch.randn(M * ldx, generator=gen)), ldx=ldx,
             w=Ref(torch.randn(cout * taps * cin, generator=gen) * (taps * cin) ** -0.5),
             in_scale=Ref(torch.rand(cin, generator=gen) + 0.5), in_shift=Ref(torch.randn(cin, generator=gen) * 0.3),
             in_relu=relu, in_mask=None, in_mask_ld=0, bias=Ref(torch.randn(cout, generator=gen)) if bias else None,
             out_mask=None, out_mask_ld=0, y=Ref(torch.zeros(M * ldy)), ldy=ldy, y_f32=1, stat_acc=None, epi_relu=0, ref=None,
             ldref=0, e_scale=None, e_shift=None, e_mean=None, e_rstd=None, s_acc=None, g=None, ldg=0, g_accumulate=0,
             g_bf16=0, dtype=1)
    out = run_both('conv', f)
    close(out['y'][0], out['y'][1], 2e-6, 'conv(fp32).y')


def _check(ref, mod, x):
    ref.eval()
    mod = mod.cuda().eval().set_compute_dtype(torch.float32)
    with torch.no_grad():
        yr = ref(x)
        ym = mod(x.cuda()).cpu()
        am = mod.predict(x.cuda()).cpu()
    err = float((yr - ym).abs().max() / yr.abs().max())
    assert err < FP32_TOL, err
    assert float((am == yr.argmax(1)).float().mean()) > 0.999
    return err


def test_tiny_densenet_fp32_logits():
    ref, mod = make_pair(TINY_CFG)
    _check(ref, mod, make_batch(2, 64, 96)[0])


def test_densenet121_fp32_logits():
    from cfgs import DenseASPP121
    ref, mod = make_pair(DenseASPP121.Model_CFG)
    err = _check(ref, mod, make_batch(1, 128, 256)[0])
    print('DenseASPP121 fp32 logits: max abs err / max |ref| = %.2e' % err)


def test_mobilenet_fp32_logits():
    ref, mod = make_mobilenet_pair()
    err = _check(ref, mod, make_batch(2, 64, 128)[0])
    print('MobileNetDenseASPP fp32 logits: max abs err / max |ref| = %.2e' % err)


def test_fp32_training_is_refused_and_bf16_comes_back():
    ref, mod = make_pair(TINY_CFG)
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    x, labels = make_batch(1, 32, 32)
    with pytest.raises(NotImplementedError):
        mod.train_step(x.cuda(), labels.cuda())
    mod.set_compute_dtype(torch.bfloat16)
    loss = mod.train_step(x.cuda(), labels.cuda())
    assert torch.isfinite(loss).all()
