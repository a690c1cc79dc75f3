"""-m gpu: evaluation counts and input pre-processing kernels against their CPU restatements (bit-exact:
integer / index work, and fp32 arithmetic with the reference's operation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_common import TINY_CFG, make_batch, make_pair  # noqa: E402
from oracle import eval_ref  # noqa: E402


def test_confusion_matrix_and_iou_match_reference_formula():
    from semantic_segmentation_amd import eval_ops
    ref, mod = make_pair(TINY_CFG)
    mod = mod.cuda().eval()
    conf = None
    preds, gts = [], []
    for seed, (B, H, W) in enumerate([(2, 64, 96), (1, 32, 48), (3, 64, 96)]):
        x, labels = make_batch(B, H, W, seed=seed + 3)
        labels[0, :4] = 7 if seed == 1 else labels[0, :4]
        conf, am = eval_ops.confusion_matrix(mod, x.cuda(), labels.cuda(), conf=conf, return_argmax=True)
        assert torch.equal(am, mod.predict(x.cuda()))          # same arg-max as the inference entry point
        preds.append(am.cpu().numpy().reshape(-1))
        gts.append(labels.numpy().reshape(-1))
    p, g = np.concatenate(preds), np.concatenate(gts)
    assert np.array_equal(conf.cpu().numpy(), eval_ref.confusion(p, g))
    assert int(conf.sum()) == p.size
    got, want = eval_ops.iou_per_class(conf), eval_ref.get_iou(p, g)
    for a, b in zip(got, want):
        assert (np.isnan(a) and np.isnan(b)) or a == b, (got, want)


def test_confusion_absent_class_is_nan_and_argmax_agrees_with_oracle():
    from semantic_segmentation_amd import eval_ops
    ref, mod = make_pair(TINY_CFG)
    mod = mod.cuda().eval()
    ref.eval()
    x, labels = make_batch(2, 64, 96)
    labels[labels == 5] = 255                                   # class 5 absent from the labels
    conf = eval_ops.confusion_matrix(mod, x.cuda(), labels.cuda())
    iou = eval_ops.iou_per_class(conf)
    assert np.isnan(iou[5]) and not np.isnan(iou[4])
    with torch.no_grad():
        am_ref = ref(x).argmax(1)
    want = eval_ref.confusion(am_ref.numpy(), labels.numpy())
    # bf16 network vs fp32 oracle: counts differ only through arg-max flips at near-ties
    assert np.abs(conf.cpu().numpy() - want).sum() <= 0.02 * labels.numel()


@pytest.mark.parametrize('hflip', [False, True])
@pytest.mark.parametrize('shape', [(2, 8, 12), (1, 33, 64), (3, 16, 4)])
def test_preprocess_bit_exact(shape, hflip):
    from semantic_segmentation_amd import eval_ops
    from semantic_segmentation_amd.backend import HipBackend
    N, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    img = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    ids = torch.randint(0, 34, (N, H, W), generator=g, dtype=torch.uint8)
    x, labels = eval_ops.preprocess(HipBackend(), img.cuda(), ids.cuda(), hflip=hflip)
    xr, lr = eval_ref.preprocess(img.numpy(), ids.numpy(), hflip=hflip)
    assert np.array_equal(x.cpu().numpy(), xr)
    assert np.array_equal(labels.cpu().numpy(), lr)
    x2, none = eval_ops.preprocess(HipBackend(), img.cuda(), None, hflip=hflip)
    assert none is None and torch.equal(x2, x)


def test_preprocess_rejects_bad_width():
    from semantic_segmentation_amd import _lib, eval_ops
    from semantic_segmentation_amd.backend import HipBackend
    with pytest.raises(_lib.DsasppError):
        eval_ops.preprocess(HipBackend(), torch.zeros(1, 4, 6, 3, dtype=torch.uint8, device='cuda'))
