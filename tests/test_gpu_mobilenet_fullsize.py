"""-m gpu: MobileNetDenseASPP (reference cfgs/MobileNetDenseASPP.py:1-8) at the size it is benchmarked at, bf16
inference of a batch of 16 images of 1024 x 2048, through properties that do not need the oracle at that size, plus
the depthwise kernel on the full 128 x 256 map of the dilated blocks against the CPU kernel model.

  * the tiled depthwise kernel (residue-class tiling of the dilated 3x3) is the one that runs, on every depthwise shape
    of the trunk at that size, and reproduces the kernel model on a 128 x 256 map with dilation 2 and 4;
  * eval logits are bit-identical for the last image alone and inside the batch of 16, and across repeated runs;
  * the three output forms (full-resolution logits, fused arg-max, confusion-matrix arg-max) agree, confusion counts are
    conserved;
  * fp32 form and bf16 form against the oracle at 1 x 1024 x 2048.

bf16 tolerance of this variant, as numbers: root-mean-square logit error <= 0.25 of the root-mean-square logit at
1024 x 2048 (measured 0.16), <= 0.2 at 64 x 128 (measured 0.12, tests/test_gpu_mobilenet.py).  Why not the DenseNet
variants' 2e-2: rounding ONLY the oracle's own conv weights and its input to bf16 already moves its logits by 12 % at
this size -- a randomly initialised 52-convolution MobileNetV2 trunk amplifies a perturbation 1.2-1.5x per block, there
is no dense connectivity to carry early features forward.  The fp32 form of the same kernels is within 1e-4 at this size.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

import kernel_model  # noqa: E402,F401
from optools import Ref, bf, close, run_both  # noqa: E402

H, W = 1024, 2048
MOBILENET_BF16_FULLSIZE_TOL = 0.25


def _lib():
    import sys
    return sys.modules['semantic_segmentation_amd']._lib


@pytest.fixture(scope='module')
def mobilenet():
    import denseASPP
    from cfgs import MobileNetDenseASPP
    torch.manual_seed(0)
    m = denseASPP.DenseASPP(MobileNetDenseASPP.Model_CFG, n_class=19, output_stride=8).cuda().eval()
    for mod in m.modules():                                    # non-trivial BatchNorm tables
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
    with torch.no_grad():
        m.classification[1].weight.mul_(8.0)                   # clear arg-max margins
    yield m
    m._engines.clear()
    torch.cuda.empty_cache()


def _batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g).clamp_(-2.2, 2.7)
    labels = torch.randint(0, 19, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return x.cuda(), labels.cuda()


def test_every_depthwise_shape_of_the_trunk_takes_the_tiled_kernel(mobilenet):
    """Each depthwise layer of the trunk at 16 x 1024 x 2048, launched alone through the C ABI: route and finiteness."""
    from semantic_segmentation_amd.backend import HipBackend
    be = HipBackend()
    eng = mobilenet.engine(16, H, W, False)
    seen = set()
    for kind, f in eng._fwd.keepalive:
        if kind != 'dwconv':
            continue
        key = (f['H'], f['W'], f['C'], f['stride'], f['dil'])
        if key in seen:
            continue
        seen.add(key)
        be.call('dwconv', **f)
        assert _lib().conv_last_route() == 'dwconv_tile', key
    torch.cuda.synchronize()
    dils = {k[4] for k in seen}
    assert {1, 2, 4} <= dils, seen                            # strided blocks made dilated past output stride 8
    assert any(k[0] == 128 and k[1] == 256 and k[4] == 4 for k in seen), seen


@pytest.mark.parametrize('C,dil', [(384, 2), (576, 2), (960, 4)])
def test_tiled_depthwise_on_the_full_map_matches_the_kernel_model(C, dil):
    """128 x 256 map (1/8 of 1024 x 2048), two images, dilation 2 / 4: residue classes, tile edges and image borders of
    the real geometry against the CPU kernel model."""
    n, h, w = 2, 128, 256
    gen = torch.Generator().manual_seed(C + dil)
    f = dict(n_img=n, H=h, W=w, C=C, stride=1, dil=dil, x=Ref(bf(torch.randn(n * h * w * C, generator=gen) * 2)), ldx=C,
             in_scale=Ref(torch.rand(C, generator=gen) + 0.5), in_shift=Ref(torch.randn(C, generator=gen) * 0.5), in_relu=2,
             w=Ref(bf(torch.randn(C * 9, generator=gen) / 3)), y=Ref(torch.zeros(n * h * w * C, dtype=torch.bfloat16)), ldy=C)
    out = run_both('dwconv', f)
    assert _lib().conv_last_route() == 'dwconv_tile'
    close(out['y'][0], out['y'][1], 1e-2, 'dwconv.y')
    # the last image row and column (tile edge + zero padding of the dilated halo)
    yc, yg = out['y'][0].view(n, h, w, C), out['y'][1].view(n, h, w, C)
    close(yc[:, -dil:], yg[:, -dil:], 1e-2, 'dwconv.y last rows')
    close(yc[:, :, -dil:], yg[:, :, -dil:], 1e-2, 'dwconv.y last columns')


def test_batch_of_16_is_batch_independent_and_deterministic(mobilenet):
    m = mobilenet
    x, _ = _batch(16, 41)
    with torch.no_grad():
        both = m.forward_lowres(x)
        again = m.forward_lowres(x)
        last = m.forward_lowres(x[15:])
        first = m.forward_lowres(x[:1])
    assert torch.isfinite(both).all() and float(both.abs().max()) > 0
    assert torch.equal(both, again)
    assert torch.equal(both[15:], last) and torch.equal(both[:1], first)


def test_output_forms_agree_and_counts_are_conserved(mobilenet):
    from semantic_segmentation_amd import eval_ops
    m = mobilenet
    x, labels = _batch(2, 42)
    with torch.no_grad():
        full = m(x)
        am = m.predict(x)
        conf, am2 = eval_ops.confusion_matrix(m, x, labels, return_argmax=True)
    assert full.shape == (2, 19, H, W) and am.shape == (2, H, W)
    assert torch.equal(am, am2)
    top2 = full.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * float(full.abs().max())
    assert bool((am == full.argmax(1))[clear].all()) and float(clear.float().mean()) > 0.99
    assert int(conf.sum()) == 2 * H * W
    assert int(conf[19].sum()) == int((labels == 255).sum())
    assert torch.equal(conf[:19].sum(1), torch.bincount(labels[labels != 255].flatten(), minlength=19))
    assert torch.equal(conf.sum(0), torch.bincount(am.flatten(), minlength=19))


def test_full_size_against_the_oracle():
    """1 x 1024 x 2048 against the fp32 oracle (about a second on the host cores for this small network).

    fp32 form of the kernels, randomised BatchNorm tables: within 1e-4 -- planner and kernels are right at this size.
    bf16 default, BatchNorm tables as constructed: root-mean-square error within MOBILENET_BF16_FULLSIZE_TOL of the
    root-mean-square logit (measured 0.16), and within 1.6x of what the ORACLE ITSELF moves when only its conv weights and
    input are rounded to bf16 (measured 0.12 at this size; the ratio is 1.25-1.4 at every size tried).  The error of
    this variant grows with the image: at 64 x 128 most taps of the dilated depthwise layers read padding."""
    from e2e_common import make_batch
    from test_mobilenet_cpu import bf16_yardstick, make_mobilenet_pair, rms_rel
    torch.set_num_threads(16)
    x, _ = make_batch(1, H, W)
    # ---- fp32 form
    ref, mod = make_mobilenet_pair()
    ref.eval()
    with torch.no_grad():
        yr = ref.forward_lowres(x)
    mod = mod.cuda().eval().set_compute_dtype(torch.float32)
    with torch.no_grad():
        y32 = mod.forward_lowres(x.cuda()).cpu()
    e32 = rms_rel(yr, y32)
    print('MobileNetDenseASPP 1x1024x2048 fp32 form vs oracle: rms %.3e, max|d|/max|ref| %.3e'
          % (e32, float((yr - y32).abs().max() / yr.abs().max())))
    assert e32 < 1e-4                                    # measured 3.6e-5
    assert float((yr - y32).abs().max() / yr.abs().max()) < 1e-3
    mod._engines.clear()
    del mod
    torch.cuda.empty_cache()
    # ---- bf16 default
    ref, mod = make_mobilenet_pair()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_parameters()
    mod.load_state_dict(ref.state_dict())
    ref.eval()
    with torch.no_grad():
        yr = ref.forward_lowres(x)
    yard = bf16_yardstick(ref, x)
    mod = mod.cuda().eval()
    with torch.no_grad():
        y16 = mod.forward_lowres(x.cuda()).cpu()
    e16 = rms_rel(yr, y16)
    print('MobileNetDenseASPP 1x1024x2048 bf16 vs oracle: rms %.3e (oracle with bf16 weights and input: %.3e)' % (e16, yard))
    assert e16 < MOBILENET_BF16_FULLSIZE_TOL
    assert e16 < 1.6 * yard
    # the bottom-right corner of the map lies at the far end of every buffer
    corner = float((yr[..., -8:, -8:] - y16[..., -8:, -8:]).pow(2).mean().sqrt() / yr.pow(2).mean().sqrt())
    assert corner < 2 * MOBILENET_BF16_FULLSIZE_TOL
    mod._engines.clear()
    torch.cuda.empty_cache()
