"""CPU tests (no GPU) of the MobileNetDenseASPP variant: module surface and the recorded eval-mode program
run through the CPU kernel model against the oracle (oracle/mobilenet_dense_aspp_ref.py)."""
import pytest
import torch

import denseASPP
import kernel_model
from cfgs import MobileNetDenseASPP
from e2e_common import make_batch
from oracle import mobilenet_dense_aspp_ref as mref

CFG = MobileNetDenseASPP.Model_CFG


def make_mobilenet_pair(seed=0):
    torch.manual_seed(seed)
    ref = mref.MobileNetDenseASPPRef(CFG)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    mod = denseASPP.DenseASPP(CFG)
    mod.load_state_dict(ref.state_dict())
    return ref, mod


@pytest.fixture
def fp32_model():
    kernel_model.set_act_dtype(torch.float32)
    yield
    kernel_model.set_act_dtype(torch.bfloat16)


def test_state_dict_and_macs_match_oracle():
    ref, mod = make_mobilenet_pair()
    sr, sm = ref.state_dict(), mod.state_dict()
    assert list(sr.keys()) == list(sm.keys())
    for k in sr:
        assert sr[k].shape == sm[k].shape and torch.equal(sr[k], sm[k]), k
    assert mod.spec.kind == 'mobilenet'
    assert mod.conv_macs(1024, 2048) == mref.conv_macs(CFG, 1024, 2048)
    # both constructor forms of the reference call sites (graph_seg_test.py:129)
    assert denseASPP.DenseASPP(object(), model_cfg=CFG).spec.kind == 'mobilenet'


def test_eval_forward_fp32(fp32_model):
    ref, mod = make_mobilenet_pair()
    mod.set_backend(kernel_model.ModelBackend())
    x, _ = make_batch(2, 64, 96)
    ref.eval(); mod.eval()
    with torch.no_grad():
        yr, ym = ref(x), mod(x)
    assert ym.shape == yr.shape == (2, 19, 64, 96)
    assert float((yr - ym).abs().max() / yr.abs().max()) < 1e-4
    assert float((mod.predict(x) == yr.argmax(1)).float().mean()) > 0.999


def bf16_yardstick(ref, x):
    """Error of the fp32 oracle against itself when only its conv weights and the input are rounded to bf16.

    A randomly initialised 52-convolution MobileNetV2 trunk amplifies perturbations by 1.2-1.5x per block
    (no dense connectivity to carry early features forward), so the 2e-2 budget that holds for the DenseNet
    variant is not meaningful here; the product's bf16 error is bounded by a multiple of this inherent one."""
    import copy
    r2 = copy.deepcopy(ref)
    for p in r2.parameters():
        if p.dim() == 4:
            p.data = p.data.bfloat16().float()
    with torch.no_grad():
        yr, y2 = ref.forward_lowres(x), r2.forward_lowres(x.bfloat16().float())
    return float((yr - y2).pow(2).mean().sqrt() / yr.pow(2).mean().sqrt())


def rms_rel(yr, y):
    return float((yr - y).pow(2).mean().sqrt() / yr.pow(2).mean().sqrt())


def test_eval_forward_bf16_model():
    ref, mod = make_mobilenet_pair()
    mod.set_backend(kernel_model.ModelBackend())
    x, _ = make_batch(1, 64, 128)
    ref.eval(); mod.eval()
    with torch.no_grad():
        yr, ym = ref.forward_lowres(x), mod.forward_lowres(x)
    yard = bf16_yardstick(ref, x)
    assert rms_rel(yr, ym) < 2.0 * yard, (rms_rel(yr, ym), yard)


def test_train_step_fp32(fp32_model):
    """Forward + backward of the whole MobileNetV2 variant (inverted residual blocks with stride 1/2 and
    dilation 1/2/4, skip connections, ReLU6 masks, BatchNorm in training mode) against the oracle's autograd.

    The loss must agree to fp32 accuracy.  The gradients of this randomly initialised 52-convolution network
    are sensitive to rounding (ReLU6 mask flips, small-batch BatchNorm): the fp32 oracle itself is 0.3 % away
    from the same oracle in fp64, so the bar is set against the fp64 oracle with the fp32 oracle's own
    deviation as the yardstick."""
    import copy
    from e2e_common import grad_report, oracle_train
    from oracle.dense_aspp_ref import make_dropout_masks
    ref, mod = make_mobilenet_pair()
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(2, 64, 96)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    ref64 = copy.deepcopy(ref).double()
    _, g64 = oracle_train(ref64, x.double(), labels, {k: v.double() for k, v in masks.items()})
    g64 = {k: v.float() for k, v in g64.items()}
    loss_r, g32 = oracle_train(ref, x, labels, masks)
    yard, _, _ = grad_report(g64, g32)
    mod.train()
    loss_m = mod.train_step(x, labels, dropout_masks=masks)
    assert abs(float(loss_r) - float(loss_m)) < 1e-5
    rel, cos, worst = grad_report(g64, {n: p.grad for n, p in mod.named_parameters()})
    assert rel < 2.0 * yard + 1e-3 and cos > 0.9999, (rel, yard, cos, worst)
    assert worst[0] < 0.03, worst
    br = dict(ref.named_buffers())
    for n, b in mod.named_buffers():
        assert float((br[n].float() - b.float()).abs().max()) < 1e-4, n


def test_input_gradient_fp32(fp32_model):
    """Gradient with respect to the image for the MobileNetV2 trunk (3x3/2 stem): planner + kernel model in fp32
    against the oracle."""
    import torch.nn.functional as F
    from oracle.dense_aspp_ref import make_dropout_masks
    ref, mod = make_mobilenet_pair()
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(2, 32, 48)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    ref.train()
    xr = x.clone().requires_grad_(True)
    F.cross_entropy(ref(xr, masks), labels, ignore_index=255).backward()
    mod.train()
    mod.train_step(x, labels, dropout_masks=masks)
    got = mod.input_gradient(x)
    rel = float((got - xr.grad).norm() / xr.grad.norm())
    assert rel < 2e-2, rel          # the fp32 oracle itself is 3e-3 from the fp64 oracle for this network's gradients
