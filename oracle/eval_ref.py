"""CPU restatement of the reference's evaluation metric and tensor pre-processing -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product package never does.
"""
import numpy as np

# reference graph_seg_dataloader.py:111-116
ID_TO_TRAINID = {-1: 255, 0: 255, 1: 255, 2: 255, 3: 255, 4: 255, 5: 255, 6: 255, 7: 0, 8: 1, 9: 255, 10: 255, 11: 2,
                 12: 3, 13: 4, 14: 255, 15: 255, 16: 255, 17: 5, 18: 255, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11,
                 25: 12, 26: 13, 27: 14, 28: 15, 29: 255, 30: 255, 31: 16, 32: 17, 33: 18}


def get_iou(pred_argmax, gt, n_class=19):
    """reference graph_seg_test.py:61-78 (and graph_seg_main.py:91-101) on an arg-max map."""
    p = np.asarray(pred_argmax).reshape(-1)
    g = np.asarray(gt).reshape(-1)
    out = []
    for c in range(n_class):
        tc, tl = p == c, g == c
        if tl.sum() == 0:
            out.append(np.nan)
        else:
            inter = float(np.logical_and(tc, tl).sum())
            union = float(np.logical_or(tc, tl).sum())
            out.append((inter + 1e-10) / (union + 1e-10))
    return out


def confusion(pred_argmax, gt, n_class=19):
    p = np.asarray(pred_argmax).reshape(-1).astype(np.int64)
    g = np.asarray(gt).reshape(-1).astype(np.int64)
    row = np.where((g >= 0) & (g < n_class), g, n_class)
    conf = np.zeros((n_class + 1, n_class), dtype=np.int64)
    np.add.at(conf, (row, p), 1)
    return conf


def preprocess(img_u8, label_ids=None, hflip=False):
    """reference graph_seg_dataloader.py:129-131 (label table), :136 (/255), :177-180 (flip), :213-214
    (normalise, as torchvision's Normalize: sub then true divide in fp32), :226-227 (CHW float / int64)."""
    img = np.asarray(img_u8)
    x = img.astype(np.float32) / np.float32(255.0)
    gt = None
    if label_ids is not None:
        ids = np.asarray(label_ids)
        gt = ids.copy()
        for k, v in ID_TO_TRAINID.items():
            if k >= 0:
                gt[ids == k] = v
        gt[ids > 33] = 255        # ids outside the table do not occur in Cityscapes; the product maps them to ignore
        gt = gt.astype(np.int64)
    if hflip:
        x = x[:, :, ::-1, :]
        if gt is not None:
            gt = gt[:, :, ::-1]
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    x = (x - mean) / std
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)), (None if gt is None else np.ascontiguousarray(gt))
