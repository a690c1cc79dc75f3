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


def test_fp32_training_is_refused_for_the_mobilenet_trunk_and_bf16_comes_back():
    ref, mod = make_mobilenet_pair()
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    x, labels = make_batch(1, 64, 64)
    with pytest.raises(NotImplementedError):
        mod.train_step(x.cuda(), labels.cuda())
    mod.set_compute_dtype(torch.bfloat16)
    loss = mod.train_step(x.cuda(), labels.cuda())
    assert torch.isfinite(loss).all()


# ------------------------------------------------------------------ fp32 training form: the gradient check that can fail
GRAD_TOL = 1e-3          # global relative L2 error of all parameter gradients against the fp32 oracle


def _train_check(cfg, B, H, W, randomize_bn=True, group=4, want_all=False):
    """fp32 storage and arithmetic on the GPU against the fp32 oracle on the CPU, same weights, inputs and dropout masks:
    loss, every parameter gradient and the BatchNorm running statistics.  Only the summation order differs."""
    from e2e_common import grad_report, make_dropout_masks, oracle_train
    ref, mod = make_pair(cfg, randomize_bn=randomize_bn)
    x, labels = make_batch(B, H, W)
    masks = make_dropout_masks(ref, B, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    mod.dgrad_group = group
    loss = mod.train_step(x.cuda(), labels.cuda(), dropout_masks={k: v.cuda() for k, v in masks.items()})
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_r)) < 1e-5 * abs(float(loss_r)), (float(loss), float(loss_r))
    g = {n: p.grad.detach().cpu() for n, p in mod.named_parameters()}
    assert all(torch.isfinite(v).all() for v in g.values())
    rel, cos, worst = grad_report(g_ref, g)
    br = dict(ref.named_buffers())
    for n, b in mod.named_buffers():
        assert float((br[n].float() - b.float().cpu()).abs().max()) < 1e-4, n
    if want_all:
        return rel, cos, worst, ref, g_ref, g, (x, labels, masks)
    return rel, cos, worst


@pytest.mark.parametrize('group', [1, 4])
def test_tiny_densenet_fp32_gradients(group):
    rel, cos, worst = _train_check(TINY_CFG, 2, 64, 96, group=group)
    print('tiny cfg fp32 gradients vs oracle: rel-L2 %.2e, cos %.7f, worst tensor %s' % (rel, cos, worst))
    assert rel < GRAD_TOL and cos > 0.99999, (rel, cos, worst)


def _group_err(g_ref, g, prefix):
    num = sum(float((g_ref[k].double() - g[k].double()).pow(2).sum()) for k in g_ref if k.startswith(prefix))
    den = sum(float(g_ref[k].double().pow(2).sum()) for k in g_ref if k.startswith(prefix))
    return (num / den) ** 0.5


def test_densenet121_fp32_gradients():
    """DenseASPP121 at 128x256.  Through ~120 stacked convolutions the fp32 gradient of a randomly initialised network
    is itself only accurate to several 1e-3: the fp32 oracle differs from the SAME oracle in fp64 by 5e-3 (relative
    L2, almost all of it in the stem and the first dense block, the far end of the backward chain).  The check
    therefore has two parts: the short end of the chain (classifier, ASPP stack, last dense block) must match the
    fp32 oracle tightly, and over all parameters the GPU must be as close to the fp64 oracle as the fp32 oracle is."""
    import copy
    from cfgs import DenseASPP121
    from e2e_common import grad_report, oracle_train
    rel, cos, worst, ref, g_ref, g, (x, labels, masks) = _train_check(DenseASPP121.Model_CFG, 1, 128, 256, randomize_bn=False,
                                                                      want_all=True)
    groups = {p: _group_err(g_ref, g, p) for p in ('classification', 'ASPP_', 'features.denseblock4', 'features.denseblock3',
                                                   'features.denseblock2', 'features.denseblock1', 'features.conv0')}
    print('DenseASPP121 128x256 fp32 gradients vs fp32 oracle: rel-L2 %.2e, cos %.7f; by group %s'
          % (rel, cos, {k: '%.1e' % v for k, v in groups.items()}))
    ref64 = copy.deepcopy(ref).double()
    # the oracle's forward updated the running statistics; gradients in train mode do not depend on them
    _, g64 = oracle_train(ref64, x.double(), labels, {k: v.double() for k, v in masks.items()})
    g64 = {k: v.float() for k, v in g64.items()}
    yard, _, _ = grad_report(g64, g_ref)
    mine, cos64, worst64 = grad_report(g64, g)
    print('against the fp64 oracle: fp32 oracle %.2e, GPU fp32 form %.2e (cos %.7f)' % (yard, mine, cos64))
    assert groups['classification'] < 1e-4 and groups['ASPP_'] < GRAD_TOL, groups
    # The far end of the chain (stem, first dense blocks) amplifies the summation-order noise of the fp32 atomics: over
    # repeated runs of the same build the GPU figure moved between 7e-3 and 1.3e-2 (fp32 oracle: 5.9e-3).  The near end
    # is held to the yardstick tightly, the whole to a multiple of it.
    for prefix in ('classification', 'ASPP_', 'features.denseblock4', 'features.denseblock3'):
        y_p, m_p = _group_err(g64, g_ref, prefix), _group_err(g64, g, prefix)
        assert m_p < 2.0 * y_p + 1e-4, (prefix, m_p, y_p)
    assert mine < 5.0 * yard + 1e-4 and mine < 3e-2 and cos64 > 0.9995, (mine, yard, cos64, worst64)


def test_bf16_default_gradients_against_the_fp32_form():
    """The default training numerics (bf16 activations and weights in the MFMA, bf16 gradient slab and BatchNorm-backward
    inputs) against the fp32 form of the same network on the same GPU, DenseASPP121 at 128x256: the tolerance of the
    default path as numbers.  Measured on MI355X: loss 2e-4 apart; classifier gradient 3.2e-2, ASPP stack 7.0e-2
    relative L2; from the last dense block backwards 0.64-0.88 (cosine 0.66 over all parameters) -- norm5's backward
    removes the per-channel mean and x-hat components of its input gradient, which at random initialisation are nearly
    all of it, and amplifies the operands' bf16 rounding by that ratio.  fp32 storage of the gradient slab or of the
    BatchNorm-backward inputs changes none of these numbers (0.82 -> 0.82): the noise comes from the bf16 MFMA
    operands, as it does for the oracle itself when only its conv weights are rounded to bf16."""
    from cfgs import DenseASPP121
    from e2e_common import grad_report, make_dropout_masks
    ref, mod = make_pair(DenseASPP121.Model_CFG, randomize_bn=False)
    x, labels = make_batch(1, 128, 256)
    masks = {k: v.cuda() for k, v in make_dropout_masks(ref, 1, torch.Generator().manual_seed(5)).items()}
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    loss32 = float(mod.train_step(x.cuda(), labels.cuda(), dropout_masks=masks))
    g32 = {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    mod.load_state_dict(ref.state_dict())              # the training step moved the running statistics
    mod.set_compute_dtype(torch.bfloat16)
    loss16 = float(mod.train_step(x.cuda(), labels.cuda(), dropout_masks=masks))
    g16 = {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    rel, cos, worst = grad_report(g32, g16)
    e_cls, e_aspp = _group_err(g32, g16, 'classification'), _group_err(g32, g16, 'ASPP_')
    print('DenseASPP121 128x256 bf16-default vs fp32-form gradients: rel-L2 %.3e, cos %.6f, classifier %.3e, ASPP %.3e, loss %.5f vs %.5f'
          % (rel, cos, e_cls, e_aspp, loss16, loss32))
    assert abs(loss16 - loss32) < 2e-3 * abs(loss32)
    assert e_cls < 0.06 and e_aspp < 0.12, (e_cls, e_aspp)
    # At random initialisation only (measured 0.82 / 0.66, stable from run to run to two digits); the tolerance of the
    # default path that one can train on is taken away from initialisation, where norm5's backward no longer cancels
    # nearly all of its input: tests/test_gpu_trajectory.py::test_gradients_away_from_initialisation (cos > 0.99).
    assert cos > 0.58 and rel < 0.92, (rel, cos, worst)


def test_input_gradient_on_gpu():
    """Gradient with respect to the image batch (reference loop with an image that requires grad, graph_seg_main.py:237-239):
    fp32 form of the kernels against the oracle; the bf16 default against the fp32 form at the head of the tolerance the
    parameter gradients of this configuration have (the image gradient sits at the far end of the backward chain)."""
    import torch.nn.functional as F
    from e2e_common import make_dropout_masks
    ref, mod = make_pair(TINY_CFG)
    x, labels = make_batch(2, 64, 96)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    ref.train()
    xr = x.clone().requires_grad_(True)
    F.cross_entropy(ref(xr, masks), labels, ignore_index=255).backward()
    want = xr.grad
    gm = {k: v.cuda() for k, v in masks.items()}
    mod = mod.cuda().train().set_compute_dtype(torch.float32)
    mod.train_step(x.cuda(), labels.cuda(), dropout_masks=gm)
    g32 = mod.input_gradient(x).cpu()
    rel32 = float((g32 - want).norm() / want.norm())
    mod.load_state_dict(ref.state_dict())
    mod.set_compute_dtype(torch.bfloat16)
    mod.train_step(x.cuda(), labels.cuda(), dropout_masks=gm)
    g16 = mod.input_gradient(x).cpu()
    rel16 = float((g16 - want).norm() / want.norm())
    cos16 = float((g16 * want).sum() / (g16.norm() * want.norm()))
    print('input gradient, tiny cfg 2x64x96: fp32 form vs oracle %.2e; bf16 vs oracle rel-L2 %.3f cos %.4f' % (rel32, rel16, cos16))
    assert rel32 < 1e-3
    assert torch.isfinite(g16).all() and rel16 < 0.8 and cos16 > 0.7
    # through autograd, as the reference's loop would see it
    mod.load_state_dict(ref.state_dict())
    xm = x.cuda().requires_grad_(True)
    F.cross_entropy(mod(xm), labels.cuda(), ignore_index=255).backward()
    assert xm.grad is not None and xm.grad.shape == xm.shape and torch.isfinite(xm.grad).all()
    assert float(xm.grad.abs().max()) > 0
