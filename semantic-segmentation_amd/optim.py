"""Fused Adam over the flat parameter arena.

Takes over torch.optim.Adam(model.parameters(), weight_decay=wd, lr=lr) with the per-epoch poly
learning-rate rule of the reference loop (graph_seg_main.py:186,241-245): one kernel updates
fp32 master weights, both moments and the bf16 shadow weights; a second rebuilds the transposed
copies the data-gradient kernels read.
"""
import ctypes

import torch

from . import _lib


class FusedAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = None
        self.v = None
        self.param_groups = [dict(lr=lr)]        # the reference loop rewrites param_group['lr'] every step

    @staticmethod
    def poly_lr(base_lr, epoch, num_epochs, power=0.9):
        return base_lr * ((1 - epoch / num_epochs) ** power)

    def zero_grad(self):
        self.model._gflat_zero()

    def step(self, grad_scale=1.0):
        mod = self.model
        mod._check_arena()
        flat = mod._flat
        if self.m is None or self.m.data_ptr() == 0 or self.m.device != flat.device or self.m.numel() != flat.numel():
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
        self.step_count += 1
        lib = _lib.load()
        d = _lib.adam_desc()
        d.n = flat.numel()
        d.p, d.g, d.m, d.v = flat.data_ptr(), mod._gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        d.w16 = mod._w16.data_ptr()
        d.lr = float(self.param_groups[0]['lr'])
        d.beta1, d.beta2 = self.betas
        d.eps, d.weight_decay = self.eps, self.weight_decay
        d.step = self.step_count
        d.grad_scale = float(grad_scale)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.dsaspp_adam_step(ctypes.byref(d), stream))
        mod._get_backend().refresh_weights(mod, cast=False)
        mod._weights_fresh = True
