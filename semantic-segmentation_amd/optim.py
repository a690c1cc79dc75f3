"""Fused Adam over the flat parameter arena.

Takes over torch.optim.Adam(model.parameters(), weight_decay=wd, lr=lr) with the per-epoch poly
learning-rate rule of the reference loop (graph_seg_main.py:186,241-245): one kernel updates
fp32 master weights, both moments and the bf16 shadow weights; a second rebuilds the transposed
copies the data-gradient kernels read.

Checkpoints: state_dict() / load_state_dict() speak torch.optim.Adam's format (per-parameter
``step`` / ``exp_avg`` / ``exp_avg_sq`` in model.parameters() order, one param group), so the
reference's ``'optimizer': optimizer.state_dict()`` / ``optimizer.load_state_dict(...)``
(graph_seg_main.py:199,280) round-trips between this optimizer and torch.optim.Adam.
"""
import ctypes

import torch

from . import _lib


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.DataParallel) else model


class FusedAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = _unwrap(model)
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.step_count = 0
        self.m = None
        self.v = None
        # the reference loop rewrites param_group['lr'] every step; the other entries mirror torch.optim.Adam's group
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)]

    @staticmethod
    def poly_lr(base_lr, epoch, num_epochs, power=0.9):
        return base_lr * ((1 - epoch / num_epochs) ** power)

    @property
    def global_step(self):
        """Optimizer steps taken so far (what the reference stores as checkpoint['global_step'])."""
        return self.step_count

    def zero_grad(self, set_to_none=False):
        self.model._gflat_zero()

    def _moments(self):
        flat = self.model._flat
        if self.m is None or self.m.data_ptr() == 0 or self.m.device != flat.device or self.m.numel() != flat.numel():
            m, v = torch.zeros_like(flat), torch.zeros_like(flat)
            if self.m is not None and self.m.numel() == flat.numel():     # the model moved: the moments follow it
                m.copy_(self.m)
                v.copy_(self.v)
            self.m, self.v = m, v

    def step(self, grad_scale=None):
        """One Adam update from the flat gradient arena.  grad_scale multiplies the gradients first; by default it is
        what the last GradReducer.run_backward left on the model (1 / world size after a summing all-reduce), else 1."""
        mod = self.model
        mod._check_arena()
        flat = mod._flat
        self._moments()
        if grad_scale is None:
            grad_scale = getattr(mod, '_grad_scale', 1.0)
        # the scale a GradReducer left belongs to the gradients of THAT backward pass: used once, then back to 1, so
        # that a later step on locally computed gradients (train_step without a reducer, loss.backward()) is not scaled
        mod._grad_scale = 1.0
        g0 = mod._first_param.grad
        if g0 is None or g0.data_ptr() != mod._gflat.data_ptr() + 4 * mod.arena.by_name[mod._param_names[0]].off:
            mod.sync_grads_to_arena()         # gradients came from loss.backward() through model(x), not from train_step
        self.step_count += 1
        lib = _lib.load()
        g = self.param_groups[0]
        d = _lib.adam_desc()
        d.n = flat.numel()
        d.p, d.g, d.m, d.v = flat.data_ptr(), mod._gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        # fp32 compute mode: the "shadow" IS the master arena (the fp32 kernels read it directly); writing packed bf16
        # there would overwrite the parameters
        shadow_is_master = mod._w16 is None or mod._w16.data_ptr() == flat.data_ptr() or mod._w16.dtype != torch.bfloat16
        d.w16 = None if shadow_is_master else mod._w16.data_ptr()
        d.lr = float(g['lr'])
        d.beta1, d.beta2 = g.get('betas', self.betas)
        d.eps, d.weight_decay = float(g.get('eps', self.eps)), float(g.get('weight_decay', self.weight_decay))
        d.step = self.step_count
        d.grad_scale = float(grad_scale)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.dsaspp_adam_step(ctypes.byref(d), stream))
        mod._get_backend().refresh_weights(mod, cast=shadow_is_master)
        mod._weights_fresh = True

    # ------------------------------------------------------------------ checkpoint format of torch.optim.Adam
    def state_dict(self):
        mod = self.model
        names = mod._param_names
        state = {}
        if self.m is not None and self.step_count > 0:
            for i, n in enumerate(names):
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=mod.arena.view(self.m, n).clone(),
                                exp_avg_sq=mod.arena.view(self.v, n).clone())
        group = dict(self.param_groups[0])
        group['params'] = list(range(len(names)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        mod = self.model
        names = mod._param_names
        groups = sd['param_groups']
        ids = [i for g in groups for i in g['params']]
        if len(ids) != len(names):
            raise ValueError('optimizer state holds %d parameters, the model has %d' % (len(ids), len(names)))
        g0 = {k: v for k, v in groups[0].items() if k != 'params'}
        for g in groups[1:]:
            for k in ('lr', 'betas', 'eps', 'weight_decay'):
                if k in g and g[k] != g0.get(k):
                    raise ValueError('FusedAdam keeps one parameter group; the loaded groups differ in %r' % k)
        self.param_groups[0].update(g0)
        self.betas = tuple(self.param_groups[0]['betas'])
        self.eps, self.weight_decay = self.param_groups[0]['eps'], self.param_groups[0]['weight_decay']
        state = sd.get('state', {})
        mod._check_arena()
        self.m = None
        self._moments()
        self.step_count = 0
        if state:
            steps = set()
            named = dict(mod.named_parameters())
            for pos, pid in enumerate(ids):
                st = state.get(pid)
                if st is None:
                    raise ValueError('optimizer state has no entry for parameter %d (%s)' % (pid, names[pos]))
                if tuple(st['exp_avg'].shape) != tuple(named[names[pos]].shape):
                    raise ValueError('optimizer state of %s has shape %s, the parameter %s'
                                     % (names[pos], tuple(st['exp_avg'].shape), tuple(named[names[pos]].shape)))
                mod.arena.view(self.m, names[pos]).copy_(st['exp_avg'])
                mod.arena.view(self.v, names[pos]).copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
            if len(steps) != 1:
                raise ValueError('FusedAdam keeps one step counter; the loaded state has steps %s' % sorted(steps))
            self.step_count = steps.pop()
