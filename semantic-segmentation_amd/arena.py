"""Flat parameter arena: every learnable tensor of the network lives in one fp32 buffer.

Layout decisions (all for the kernels' benefit):
  * convolution weights are stored [cout][kh][kw][cin] -- the K-contiguous layout the
    implicit-GEMM kernels read -- and exposed to PyTorch as the usual [cout, cin, kh, kw]
    Parameter through a permuted view (this is exactly torch.channels_last), so
    state_dict() values, shapes and keys are those of a plain nn.Conv2d;
  * every entry starts at a multiple of 8 elements (16-byte aligned in bf16);
  * the classifier is padded to 32 output rows (rows >= n_class stay zero forever);
  * gradients live in a second arena with identical offsets, so a bucketed all-reduce or
    a fused optimizer step is a walk over one contiguous buffer;
  * a bf16 shadow arena (same offsets) and a transposed bf16 arena feed the kernels.
"""
import torch


class Entry:
    __slots__ = ('name', 'shape', 'kind', 'off', 'numel', 'rows', 'cin', 'taps', 'toff', 'pad_rows')

    def __init__(self, name, shape, kind):
        self.name = name
        self.shape = tuple(shape)
        self.kind = kind        # 'conv' | 'vec' | 'stem'
        self.off = 0
        self.toff = -1
        self.pad_rows = 0


def _align(n, a=8):
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self):
        self.entries = []
        self.by_name = {}
        self.size = 0
        self.tsize = 0
        self.flat = None
        self.grad = None

    def add(self, name, shape, kind, pad_rows=0):
        e = Entry(name, shape, kind)
        e.off = self.size
        if kind in ('conv', 'stem'):
            cout, cin, kh, kw = shape
            e.rows = max(cout, pad_rows)
            e.cin = cin
            e.taps = kh * kw
            e.numel = e.rows * cin * kh * kw
            if kind == 'conv':
                e.toff = self.tsize
                self.tsize += _align(e.numel)
        else:
            e.rows = max(shape[0], pad_rows)
            e.cin = 1
            e.taps = 1
            e.numel = e.rows
        e.pad_rows = e.rows
        self.size += _align(e.numel)
        self.entries.append(e)
        self.by_name[name] = e
        return e

    def allocate(self, device):
        self.flat = torch.zeros(self.size, dtype=torch.float32, device=device)
        return self.flat

    def view(self, flat, name):
        """PyTorch-shaped view of entry `name` inside `flat` (any arena with these offsets)."""
        e = self.by_name[name]
        if e.kind in ('conv', 'stem'):
            cout, cin, kh, kw = e.shape
            n = cout * cin * kh * kw
            return flat[e.off:e.off + n].view(cout, kh, kw, cin).permute(0, 3, 1, 2)
        return flat[e.off:e.off + e.shape[0]]
