"""Host side of the two kernels next to the forward path: evaluation counts and input pre-processing.

  * confusion_matrix / iou_per_class: what the reference's evaluate_model (graph_seg_main.py:71-107) and
    get_iou (graph_seg_test.py:61-78) compute per batch -- arg-max of the full-resolution scores and, per
    class, |pred == c AND gt == c| / |pred == c OR gt == c| (NaN for classes absent from the labels) --
    from ONE fused upsample + arg-max + histogram kernel and no per-class device synchronisation.
  * preprocess: the tensor side of the data loader (graph_seg_dataloader.py:111-116,129-131,136,177-180,
    213-214,226-227) on the GPU, so a batch crosses PCIe as bytes.
"""
import math

import torch

from . import spec as S

# Cityscapes labelId -> trainId (reference graph_seg_dataloader.py:111-116); every other id, and -1, is ignored
ID_TO_TRAINID = {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13,
                 27: 14, 28: 15, 31: 16, 32: 17, 33: 18}
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # reference graph_seg_dataloader.py:213
IMAGENET_STD = (0.229, 0.224, 0.225)       # reference graph_seg_dataloader.py:214


def trainid_lut(ignore_label=255, device='cpu'):
    lut = torch.full((256,), ignore_label, dtype=torch.uint8)
    for k, v in ID_TO_TRAINID.items():
        lut[k] = v
    return lut.to(device)


def confusion_matrix(model, x, labels, conf=None, return_argmax=False):
    """Run the eval-mode forward of `model` on x and add this batch's counts to `conf`
    (int64 [(n_class+1), n_class]: row = label, row n_class = ignored / out-of-range labels; column = arg-max)."""
    nc = model.spec.n_class
    eng = model._run_forward(x)
    if eng.training:
        raise RuntimeError('confusion_matrix needs eval() mode')
    if labels.shape != (eng.B, eng.H, eng.W) or labels.dtype != torch.int64 or labels.device != x.device:
        raise ValueError('labels must be an int64 tensor of shape %s on the input device' % ((eng.B, eng.H, eng.W),))
    if conf is None:
        conf = torch.zeros(nc + 1, nc, dtype=torch.int64, device=x.device)
    am = torch.empty(eng.B, eng.H, eng.W, dtype=torch.int64, device=x.device) if return_argmax else None
    labels = labels.contiguous()
    model._get_backend().call('confusion', n_img=eng.B, h=eng.H2, w=eng.W2, n_class=nc, ldl=S.CLS_PAD,
                              logits=eng.logits.data_ptr(), labels=labels.data_ptr(), conf=conf.data_ptr(),
                              argmax=am.data_ptr() if am is not None else None)
    return (conf, am) if return_argmax else conf


def iou_per_class(conf):
    """Per-class IoU list with the reference's conventions (graph_seg_test.py:61-78): NaN when the class
    has no labelled pixel, else (intersect + 1e-10) / (union + 1e-10); one device-to-host copy."""
    c = conf.detach().cpu().double()
    nc = c.shape[1]
    out = []
    for k in range(nc):
        n_label = float(c[k].sum())
        if n_label == 0:
            out.append(math.nan)
            continue
        inter = float(c[k, k])
        union = n_label + float(c[:, k].sum()) - inter
        out.append((inter + 1e-10) / (union + 1e-10))
    return out


def preprocess(model_or_backend, img_u8, label_ids=None, hflip=False, lut=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """img_u8: uint8 [N, H, W, 3] on the GPU; label_ids: uint8 [N, H, W] Cityscapes labelIds or None.
    Returns (x fp32 [N, 3, H, W], labels int64 [N, H, W] or None)."""
    be = model_or_backend._get_backend() if hasattr(model_or_backend, '_get_backend') else model_or_backend
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
        raise ValueError('img_u8 must be a uint8 tensor of shape N x H x W x 3')
    img_u8 = img_u8.contiguous()
    N, H, W, _ = img_u8.shape
    x = torch.empty(N, 3, H, W, dtype=torch.float32, device=img_u8.device)
    labels = None
    f = dict(n_img=N, H=H, W=W, img=img_u8.data_ptr(), x=x.data_ptr(), mean0=mean[0], mean1=mean[1], mean2=mean[2],
             std0=std[0], std1=std[1], std2=std[2], hflip=int(bool(hflip)), label_ids=None, lut=None, labels=None)
    if label_ids is not None:
        if label_ids.dtype != torch.uint8 or tuple(label_ids.shape) != (N, H, W):
            raise ValueError('label_ids must be a uint8 tensor of shape N x H x W')
        label_ids = label_ids.contiguous()
        lut = trainid_lut(device=img_u8.device) if lut is None else lut
        labels = torch.empty(N, H, W, dtype=torch.int64, device=img_u8.device)
        f.update(label_ids=label_ids.data_ptr(), lut=lut.data_ptr(), labels=labels.data_ptr())
    be.call('preprocess', **f)
    return x, labels
