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
    # The bf16 tolerance of this variant against the fp32 oracle, as a number: root-mean-square logit error within 0.2 of
    # the root-mean-square logit (MOBILENET_BF16_RMS_TOL, also used at 1024x2048 in test_gpu_mobilenet_fullsize.py).
    # The yardstick printed above is why it is not the DenseNet variants' 2e-2: the oracle itself moves by that much
    # (about 9 %) when only its conv weights and input are rounded to bf16.
    assert e_oracle < 0.2, (e_oracle, yard)
    assert e_oracle < 2.5 * yard, (e_oracle, yard)
    # the full-resolution API and the fused arg-max agree with each other exactly where the top-1 margin is clear
    assert full.shape == (2, 19, 64, 128)
    top2 = full.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3 * float(full.abs().max())
    assert bool((am == full.argmax(1))[clear].all())


# ---------------------------------------------------------------- training kernels
@pytest.mark.parametrize('case', range(len(DW_CASES)))
@pytest.mark.parametrize('dx_f32', [0, 1])
def test_dwconv3x3_bwd_data(case, dx_f32):
    n, H, W, C, st, dil, relu, affine = DW_CASES[case]
    gen = torch.Generator().manual_seed(40 + case)
    ldx, lddy, lddx = C + 8, C + 16, C
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    f = dict(n_img=n, H=H, W=W, C=C, stride=st, dil=dil, x=Ref(bf(torch.randn(n * H * W * ldx, generator=gen) * 2)), ldx=ldx,
             in_scale=None, in_shift=None, in_relu=relu, w=Ref(bf(torch.randn(C * 9, generator=gen) / 3)),
             dy=Ref(bf(torch.randn(n * Ho * Wo * lddy, generator=gen))), lddy=lddy,
             dx=Ref(torch.zeros(n * H * W * lddx, dtype=torch.float32 if dx_f32 else torch.bfloat16)), lddx=lddx, dx_f32=dx_f32,
             e_mean=Ref(torch.randn(C, generator=gen) * 0.1), e_rstd=Ref(torch.rand(C, generator=gen) + 0.5),
             s_acc=Ref(torch.zeros(kernel_model.R * 2 * C)), dw=None)
    if affine:
        f['in_scale'] = Ref(torch.rand(C, generator=gen) + 0.5)
        f['in_shift'] = Ref(torch.randn(C, generator=gen) * 0.5 + 1.0)
    out = run_both('dw_bwd_data', f)
    close(out['dx'][0], out['dx'][1], 1e-2 if not dx_f32 else 1e-5, 'dw_bwd_data.dx')
    sc, sg = out['s_acc'][0].view(kernel_model.R, -1).sum(0), out['s_acc'][1].view(kernel_model.R, -1).sum(0)
    close(sc, sg, 2e-2 if not dx_f32 else 1e-4, 'dw_bwd_data.sums')


@pytest.mark.parametrize('case', range(len(DW_CASES)))
def test_dwconv3x3_wgrad(case):
    n, H, W, C, st, dil, relu, affine = DW_CASES[case]
    gen = torch.Generator().manual_seed(60 + case)
    ldx, lddy = C + 8, C
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    f = dict(n_img=n, H=H, W=W, C=C, stride=st, dil=dil, x=Ref(bf(torch.randn(n * H * W * ldx, generator=gen) * 2)), ldx=ldx,
             in_scale=None, in_shift=None, in_relu=relu, w=None,
             dy=Ref(bf(torch.randn(n * Ho * Wo * lddy, generator=gen))), lddy=lddy, dx=None, lddx=0, dx_f32=0,
             e_mean=None, e_rstd=None, s_acc=None, dw=Ref(torch.randn(C * 9, generator=gen)))     # accumulates into dw
    if affine:
        f['in_scale'] = Ref(torch.rand(C, generator=gen) + 0.5)
        f['in_shift'] = Ref(torch.randn(C, generator=gen) * 0.5 + 1.0)
    out = run_both('dw_wgrad', f)
    close(out['dw'][0], out['dw'][1], 1e-4, 'dw_wgrad.dw')


@pytest.mark.parametrize('shape', [(2, 16, 24, 32), (1, 10, 6, 40), (3, 34, 18, 32)])
def test_stem3_im2col(shape):
    n, H, W, ld = shape
    gen = torch.Generator().manual_seed(5)
    f = dict(n_img=n, H=H, W=W, x=Ref(torch.randn(n * 3 * H * W, generator=gen)),
             col=Ref(torch.full((n * (H // 2) * (W // 2) * ld,), 7.0, dtype=torch.bfloat16)), ldcol=ld)
    out = run_both('stem3_col', f)
    assert torch.equal(out['col'][0], out['col'][1])


@pytest.mark.parametrize('cin,cout', [(16, 96), (960, 160), (144, 24)])
def test_relu6_masks_in_backward_kernels(cin, cout):
    """ReLU6 in the weight-gradient prologue and as the data-gradient epilogue mask (0 < pre < 6)."""
    from test_gpu_ops import wgrad_fields
    f = wgrad_fields(2, 9, 13, cin, cout, 1, 1, seed=cin)
    f['in_relu'] = 2
    f['in_scale'] = Ref(f['in_scale'].t * 3)
    out = run_both('wgrad', f)
    close(out['dw'][0], out['dw'][1], 5e-3, 'wgrad.dw')
    for y_f32 in (False, True):
        g = conv_fields(2, 9, 13, cout, cin, 1, 1, scale=False, relu=False, y_f32=y_f32, epi_relu=True, e_scale=True,
                        s_acc=True, seed=cout)
        g['epi_relu'] = 2
        g['e_scale'] = Ref(g['e_scale'].t * 4)
        check_conv(g)


def test_mobilenet_train_step():
    """One training step on the GPU against (1) the CPU kernel model run from the identical forward state and
    (2) the fp32 oracle, with the oracle's own bf16-weight sensitivity as the yardstick."""
    import copy
    from e2e_common import grad_report, oracle_train
    from oracle.dense_aspp_ref import make_dropout_masks
    ref, mod = make_mobilenet_pair()
    cpu_mod = copy.deepcopy(mod)
    cpu_mod.set_backend(kernel_model.ModelBackend())
    B, H, W = 2, 64, 96
    x, labels = make_batch(B, H, W)
    masks = make_dropout_masks(ref, B, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod = mod.cuda().train()
    loss_g = mod.train_step(x.cuda(), labels.cuda(), dropout_masks={k: v.cuda() for k, v in masks.items()})
    torch.cuda.synchronize()
    g_gpu = {n: p.grad.cpu().clone() for n, p in mod.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g_gpu.values())

    cpu_mod.train()
    eng_c, eng_g = cpu_mod.engine(B, H, W, True), mod.engine(B, H, W, True)
    cpu_mod.refresh_weights()
    cpu_mod._gflat_zero()

    def cp(a, b):
        b.copy_(a.to(b.device))
    for k in ('x', 'mean', 'var', 'rstd'):
        cp(getattr(eng_g.aslab, k), getattr(eng_c.aslab, k))
    for k in eng_g.y1:
        cp(eng_g.y1[k], eng_c.y1[k])
    for k in eng_g.bn:
        cp(eng_g.bn[k], eng_c.bn[k])
    for kg, kc in zip(eng_g.ir_keep, eng_c.ir_keep):
        for k in ('e1', 'e2', 'p', 'out'):
            if kg[k] is not None:
                cp(kg[k], kc[k])
    for k in ('stem_out', 'x_in', 'logits', 'dropmask', 'labels'):
        cp(getattr(eng_g, k), getattr(eng_c, k))
    eng_c._fwd_loss.run()
    eng_c._bwd.run()
    cpu_mod.bind_grads()
    g_cpu = {n: p.grad.clone() for n, p in cpu_mod.named_parameters()}
    assert abs(float(eng_c.loss) - float(loss_g)) < 1e-4 * abs(float(loss_g))
    rel, cos, worst = grad_report(g_cpu, g_gpu)
    print('mobilenet train: gpu vs kernel model rel %.3e cos %.6f worst %s' % (rel, cos, worst))
    assert rel < 2e-2 and cos > 0.9995, (rel, cos, worst)

    # against the fp32 oracle: the head (classifier and ASPP stack) with a stated tolerance.  Behind the head the
    # gradient of this randomly initialised 52-convolution trunk is dominated by what its BatchNorm backward passes
    # cancel; bf16 rounding of the operands is amplified there beyond comparison (measured on MI355X: 1.2 relative L2
    # over all parameters, and 0.92 for the ORACLE ITSELF once its conv weights are rounded to bf16), so the trunk is
    # covered by (1) above and by the fp32 planner test on the CPU, not by a bf16-against-fp32 bound.
    def group(prefix):
        num = sum(float((g_ref[k] - g_gpu[k]).pow(2).sum()) for k in g_ref if k.startswith(prefix))
        return (num / sum(float(g_ref[k].pow(2).sum()) for k in g_ref if k.startswith(prefix))) ** 0.5

    rel_o, cos_o, worst_o = grad_report(g_ref, g_gpu)
    e_cls, e_aspp = group('classification'), group('ASPP_')
    print('mobilenet train: gpu vs oracle rel %.3e (all parameters), classifier %.3e, ASPP stack %.3e' % (rel_o, e_cls, e_aspp))
    assert abs(float(loss_g) - float(loss_r)) < 5e-3 * abs(float(loss_r))
    # measured: 8.0e-2 and 1.5e-1 (the forward pass of this trunk alone is 9 % off in bf16, see bf16_yardstick)
    assert e_cls < 0.12 and e_aspp < 0.25, (e_cls, e_aspp)


def test_mobilenet_trains():
    """A few optimizer steps through the reference-style loop reduce the loss."""
    from semantic_segmentation_amd.optim import FusedAdam
    _, mod = make_mobilenet_pair()
    mod = mod.cuda().train()
    x, labels = make_batch(2, 64, 96)
    x, labels = x.cuda(), labels.cuda()
    opt = FusedAdam(mod, lr=1e-3, weight_decay=1e-5)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        losses.append(float(mod.train_step(x, labels)))
        opt.step()
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
