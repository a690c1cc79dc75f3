"""DenseASPP nn.Module whose forward/backward run on the MI355X kernels of libdsaspp.so.

Drop-in surface (reference call sites): ``from denseASPP import DenseASPP``
(graph_seg_test.py:12); ``DenseASPP(args, model_cfg=cfg)`` (graph_seg_test.py:129) and the
upstream ``DenseASPP(model_cfg, n_class=19, output_stride=8)``; ``nn.DataParallel(model,
device_ids=[gpu])`` (graph_seg_test.py:131, graph_seg_main.py:173); ``state_dict`` /
``load_state_dict`` with or without the ``module.`` prefix (graph_seg_main.py:198,279,
graph_seg_test.py:135); ``train()/eval()/cuda()``; ``model(image) -> NCHW logits``
(graph_seg_main.py:237, graph_seg_test.py:144) usable with CrossEntropyLoss / softmax.

The module tree mirrors a plain PyTorch DenseASPP so that parameter and buffer names,
shapes and values are checkpoint compatible, but the child modules only *hold* tensors:
all arithmetic is recorded by engine.Engine and executed by the HIP library.  The
parameters are views into one flat fp32 arena (arena.ParamArena).
"""
import re
from collections import OrderedDict

import torch
import torch.nn as nn

from . import spec as S
from .arena import ParamArena
from .engine import Engine

_LEGACY = re.compile(r'\.(norm|relu|conv)\.(\d)\.')


def _holder_layer(cin, mid, growth, dil):
    m = nn.Module()
    m.norm1 = nn.BatchNorm2d(cin)
    m.relu1 = nn.ReLU(inplace=True)
    m.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
    m.norm2 = nn.BatchNorm2d(mid)
    m.relu2 = nn.ReLU(inplace=True)
    m.conv2 = nn.Conv2d(mid, growth, 3, dilation=dil, padding=dil, bias=False)
    return m


def _holder_aspp(cin, d0, d1, rate, bn_start):
    m = nn.Module()
    if bn_start:
        m.norm1 = nn.BatchNorm2d(cin, momentum=0.0003)
    m.relu1 = nn.ReLU(inplace=True)
    m.conv1 = nn.Conv2d(cin, d0, 1)
    m.norm2 = nn.BatchNorm2d(d0, momentum=0.0003)
    m.relu2 = nn.ReLU(inplace=True)
    m.conv2 = nn.Conv2d(d0, d1, 3, dilation=rate, padding=rate)
    return m


def _holder_ir(b):
    """Inverted-residual block: Sequential indices 0/1 expand conv+BN, 3/4 depthwise conv+BN, 6/7 linear conv+BN."""
    m = nn.Module()
    m.conv = nn.Sequential(
        nn.Conv2d(b.cin, b.hid, 1, bias=False), nn.BatchNorm2d(b.hid), nn.ReLU6(inplace=True),
        nn.Conv2d(b.hid, b.hid, 3, b.stride, padding=b.dil, dilation=b.dil, groups=b.hid, bias=False),
        nn.BatchNorm2d(b.hid), nn.ReLU6(inplace=True),
        nn.Conv2d(b.hid, b.cout, 1, bias=False), nn.BatchNorm2d(b.cout))
    return m


def _holder_mobilenet(sp):
    """MobileNetV2 trunk; parameter names features.features.<i>.* (trunk module holding a Sequential)."""
    trunk = nn.Module()
    layers = [nn.Sequential(nn.Conv2d(3, sp.c_stem, 3, 2, 1, bias=False), nn.BatchNorm2d(sp.c_stem), nn.ReLU6(inplace=True))]
    layers += [_holder_ir(b) for b in sp.ir]
    trunk.features = nn.Sequential(*layers)
    return trunk


class _LogitsFn(torch.autograd.Function):
    """Autograd bridge for the plain ``model(x)`` API (full-resolution logits)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._run_forward(x)
        eng.prog_up_fwd().run()
        ctx.model, ctx.eng = model, eng
        sp = model.spec
        return eng.out_logits.view(eng.B, sp.n_class, eng.H, eng.W).clone()

    @staticmethod
    def backward(ctx, grad):
        model, eng = ctx.model, ctx.eng
        if not eng.training:
            raise RuntimeError('DenseASPP: backward needs the forward pass to run in train() mode')
        prog = eng.prog_up_bwd()
        eng.grad_full.view(grad.shape).copy_(grad)
        # The backward program writes into the flat gradient arena, which the .grad views of a previous train_step
        # (or bind_grads) alias: keep its contents, compute this pass's gradients alone, hand them to autograd as a
        # private copy (autograd does the accumulation into .grad) and put the arena back as it was.
        held = model._gflat.clone()
        model._gflat_zero()
        prog.run()
        eng._bwd.run()
        g = model._gflat.clone()
        model._gflat.copy_(held)
        grads = [model.arena.view(g, n) for n in model._param_names]
        gx = eng.input_gradient().clone() if ctx.needs_input_grad[1] else None
        return (None, gx) + tuple(grads)


class DenseASPP(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        model_cfg, n_class, output_stride = self._parse_ctor(args, kwargs)
        sp = S.NetSpec(model_cfg, n_class, output_stride)
        self.spec = sp
        cs = sp.c_stem

        feats = OrderedDict()
        if sp.kind == 'mobilenet':
            self.features = _holder_mobilenet(sp)
        else:
            feats['conv0'] = nn.Conv2d(3, cs, 7, stride=2, padding=3, bias=False)
            feats['norm0'] = nn.BatchNorm2d(cs)
            feats['relu0'] = nn.ReLU(inplace=True)
            feats['pool0'] = nn.MaxPool2d(3, stride=2, padding=1)
            self.features = nn.Sequential(feats)
        for blk in sp.blocks:
            block = nn.Sequential()
            for i, l in enumerate(blk.layers):
                block.add_module('denselayer%d' % (i + 1), _holder_layer(l.cin, l.mid, l.growth, l.dil))
            self.features.add_module('denseblock%d' % blk.index, block)
            tr = nn.Sequential()
            tr.add_module('norm', nn.BatchNorm2d(blk.C))
            tr.add_module('relu', nn.ReLU(inplace=True))
            tr.add_module('conv', nn.Conv2d(blk.C, blk.cout, 1, bias=False))
            if blk.pool_after:
                tr.add_module('pool', nn.AvgPool2d(2, stride=2))
            self.features.add_module('transition%d' % blk.index, tr)
        if sp.kind != 'mobilenet':
            self.features.add_module('norm5', nn.BatchNorm2d(sp.c_feat))
            if sp.output_stride == 16:
                self.features.add_module('upsample', nn.Upsample(scale_factor=2, mode='bilinear'))
        for a in sp.aspp:
            setattr(self, a.prefix, _holder_aspp(a.cin, a.d0, a.d1, a.rate, a.bn_start))
        self.classification = nn.Sequential(
            nn.Dropout2d(p=sp.dropout1),
            nn.Conv2d(sp.C_aspp, n_class, 1),
            nn.Upsample(scale_factor=8, mode='bilinear'))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight.data)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

        # ---- arena bookkeeping
        self.arena = ParamArena()
        self._param_names = []
        self.bn_modules = {}
        for name, p in self.named_parameters():
            self._param_names.append(name)
            if p.dim() == 4:
                if name == 'features.conv0.weight':
                    self.arena.add(name, p.shape, 'stem')
                elif name == 'classification.1.weight':
                    self.arena.add(name, p.shape, 'conv', pad_rows=S.CLS_PAD)
                else:
                    self.arena.add(name, p.shape, 'conv')
            elif name == 'classification.1.bias':
                self.arena.add(name, p.shape, 'vec', pad_rows=S.CLS_PAD)
            else:
                self.arena.add(name, p.shape, 'vec')
        for name, m in self.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                self.bn_modules[name] = m
        self.compute_dtype = torch.bfloat16      # activation storage / MFMA input type; see set_compute_dtype
        self.grad_slab_dtype = torch.bfloat16    # dtype of the feature-slab gradient (torch.float32: exact accumulation, 2x traffic)
        self.bn_grad_dtype = torch.bfloat16      # dtype of the BatchNorm-backward inputs (torch.float32: unrounded)
        self.dgrad_group = 4                     # dense layers whose slab data gradients share one pass (1..4)
        self.wgrad_group = True                  # ... and whose bottleneck weight gradients share one pass over the slab
        # dense-layer weight gradients on a second HIP stream (per-layer gradient buffers, ~20 GB at batch 16):
        # +1.9 % step throughput measured, but overlapped kernels time longer individually, so it is off by default
        self.side_wgrad = False
        # dense-layer 3x3 weight gradients next to the 3x3 data gradients on a second stream (shared buffers, joins before
        # every overwrite): measured 275.1 / 276.7 against 276.7 / 278.4 ms per step, i.e. within run-to-run drift: off
        self.side_wgrad3 = False
        self.fuse_extract3 = True                # read a dense layer's gradient out of the slab inside its 3x3 data gradient
        self.max_engines = 4                     # (batch, height, width, mode) configurations kept with their buffers
        self._flat = None
        self._gflat = None
        self._w16 = None
        self._wt16 = None
        self._w0_16 = None
        self._wt_table = None
        self._buffers_flat = {}
        self._engines = {}
        self._backend = None
        self._flatten()
        self._register_load_state_dict_pre_hook(self._remap_keys)

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def _parse_ctor(args, kwargs):
        """Accept DenseASPP(model_cfg, n_class=19, output_stride=8) and DenseASPP(args, model_cfg=cfg)."""
        args = list(args)
        cfg = kwargs.pop('model_cfg', None)
        if cfg is None:
            if not args or not isinstance(args[0], dict):
                raise TypeError('DenseASPP needs a model_cfg dict (positional, or model_cfg=...)')
            cfg = args.pop(0)
        elif args and not isinstance(args[0], dict):
            args.pop(0)        # the reference passes its argparse namespace first; nothing in it is used
        n_class = kwargs.pop('n_class', args.pop(0) if args else 19)
        output_stride = kwargs.pop('output_stride', args.pop(0) if args else 8)
        if kwargs or args:
            raise TypeError('DenseASPP: unexpected arguments %r %r' % (args, kwargs))
        return cfg, int(n_class), int(output_stride)

    @staticmethod
    def _remap_keys(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Accept checkpoints saved from a DataParallel wrapper and legacy dotted child names."""
        for k in list(state_dict.keys()):
            nk = k
            if prefix == '' and nk.startswith('module.'):
                nk = nk[len('module.'):]
            # the key as this module names it: under an nn.DataParallel wrapper the hook runs with prefix 'module.'
            rel = nk[len(prefix):] if prefix and nk.startswith(prefix) else nk
            if '.conv.' not in rel or 'denselayer' in rel or rel.startswith('ASPP_'):
                # legacy dotted child names of dense layers / ASPP blocks ("norm.1" -> "norm1"); the MobileNet
                # trunk's "features.features.<i>.conv.<j>" names are Sequential indices and stay as they are
                nk = _LEGACY.sub(lambda m: '.%s%s.' % (m.group(1), m.group(2)), nk)
            if nk != k:
                state_dict[nk] = state_dict.pop(k)

    def _flatten(self):
        """(Re)build the flat arenas on the device the parameters currently live on."""
        dev = next(self.parameters()).device
        for p in self.parameters():
            if p.dtype != torch.float32:
                raise TypeError('DenseASPP keeps fp32 master parameters; the kernels compute in bf16 with fp32 accumulation')
        flat = self.arena.allocate(dev)
        named = dict(self.named_parameters())
        for name in self._param_names:
            v = self.arena.view(flat, name)
            v.copy_(named[name].data)
            named[name].data = v
        self._flat = flat
        object.__setattr__(self, '_first_param', named[self._param_names[0]])    # plain attribute, not a registered Parameter
        self._gflat = torch.zeros_like(flat)
        self._alloc_shadows()
        self._buffers_flat = {n: b for n, b in self.named_buffers()}
        self._engines = {}
        self._wt_table = None
        for name in self._param_names:
            named[name].grad = None

    def _alloc_shadows(self):
        dt = self._backend.act_dtype if self._backend is not None else self.compute_dtype
        dev = self._flat.device
        if dt == torch.float32 and getattr(self._backend, 'forward_only_fp32', False):
            self._w16 = self._flat      # fp32 kernels read the master weights themselves
        else:
            self._w16 = torch.zeros(self.arena.size, dtype=dt, device=dev)
        self._wt16 = torch.zeros(max(self.arena.tsize, 8), dtype=dt, device=dev)
        self._w0_16 = torch.zeros(self.spec.c_stem * S.STEM_KPAD, dtype=dt, device=dev)   # padded 7x7 stem rows (DenseNet trunk)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._flatten()
        return self

    def _check_arena(self):
        first = self._param_names[0]
        p = dict(self.named_parameters())[first] if self._flat is None else self._first_param
        if (self._flat is None or p.device != self._flat.device
                or p.data_ptr() != self._flat.data_ptr() + 4 * self.arena.by_name[first].off
                or self.classification[1].bias.data_ptr() != self._flat.data_ptr() + 4 * self.arena.by_name['classification.1.bias'].off):
            self._flatten()

    # ------------------------------------------------------------------ nn.DataParallel over several devices
    # nn.DataParallel(model) without device_ids (graph_seg_main.py:173) replicates the module on every visible GPU each
    # forward: torch broadcasts parameters and buffers and builds shallow module copies whose attributes are those
    # broadcast tensors.  The kernels need a flat arena, engines and activation buffers per device, so every device
    # other than the master's gets one persistent worker model (built on first use); a replica's forward copies the
    # broadcast values into its worker and runs the worker's programs, with the broadcast copies as the autograd inputs
    # so that torch's own reduce-add brings the replica gradients back to the master parameters.  BatchNorm running
    # statistics of the other devices are dropped, as with plain nn.BatchNorm2d under nn.DataParallel.  One process
    # per GPU (semantic_segmentation_amd.distributed) remains the fast path: this one re-broadcasts 4 bytes per
    # parameter per device and step.
    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica.__dict__['_dp_master'] = self.__dict__.get('_dp_master', self)
        return replica

    def _dp_worker(self, device):
        """The model that runs a replica's work on `device`: the master itself on its own device."""
        self._check_arena()
        if device == self._flat.device and not getattr(self, '_dp_force_worker', False):
            return self
        workers = self.__dict__.setdefault('_dp_workers', {})
        w = workers.get(device)
        if w is None:
            if not getattr(self, '_dp_warned', False):
                import warnings
                warnings.warn('DenseASPP under multi-device nn.DataParallel: every extra device keeps a full worker copy and '
                              'parameters are re-broadcast each step; one process per GPU with '
                              'semantic_segmentation_amd.distributed.GradReducer is the fast path', stacklevel=3)
                self._dp_warned = True
            w = DenseASPP(self.spec.cfg, n_class=self.spec.n_class, output_stride=self.spec.output_stride).to(device)
            for k in ('compute_dtype', 'grad_slab_dtype', 'bn_grad_dtype', 'dgrad_group', 'wgrad_group'):
                setattr(w, k, getattr(self, k))
            if self._backend is not None and not hasattr(self._backend, 'lib'):
                w.set_backend(type(self._backend)())        # test backends (CPU kernel model) follow the master
            workers[device] = w
        return w

    def _replica_forward(self, x):
        master = self._dp_master
        worker = master._dp_worker(x.device)
        copies = []
        for name in master._param_names:
            obj = self
            for part in name.split('.'):
                obj = getattr(obj, part)
            copies.append(obj)
        worker.train(self.training)
        if worker is not master:
            with torch.no_grad():
                for name, c in zip(master._param_names, copies):
                    worker.arena.view(worker._flat, name).copy_(c)
                mine = dict(self.named_buffers())
                for name, b in worker.named_buffers():
                    b.copy_(mine[name])
        if torch.is_grad_enabled() and self.training:
            return _LogitsFn.apply(worker, x, *copies)
        return worker.forward(x)

    def bind_grads(self):
        """Point every Parameter's .grad at its slice of the flat gradient arena."""
        named = dict(self.named_parameters())
        for name in self._param_names:
            named[name].grad = self.arena.view(self._gflat, name)

    def _gflat_zero(self):
        self._gflat.zero_()

    def sync_grads_to_arena(self):
        """Make the flat gradient arena hold what the Parameters' .grad hold.  After train_step the .grad tensors ARE
        views of the arena; after loss.backward() through model(x) autograd may have installed tensors of its own
        (first backward) -- those are copied in and the views re-bound, parameters without a gradient count as zero."""
        named = dict(self.named_parameters())
        for name in self._param_names:
            p = named[name]
            view = self.arena.view(self._gflat, name)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view

    # ------------------------------------------------------------------ backend / engines
    def _get_backend(self):
        if self._backend is None:
            from .backend import HipBackend
            self._backend = HipBackend(self.compute_dtype)
            self._alloc_shadows()
        return self._backend

    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (default: bf16 storage, bf16 MFMA, fp32 accumulation) or torch.float32 (fp32 storage
        and arithmetic, eval-mode inference only: the form used to check the network against an fp32
        reference to 1e-4)."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError('compute dtype must be torch.bfloat16 or torch.float32')
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._backend = None
            self._engines = {}
            self._wt_table = None
        return self

    def set_backend(self, backend):
        """Test hook: run descriptor lists through another interpreter (CPU kernel model in tests/)."""
        self._backend = backend
        self._engines = {}
        self._alloc_shadows()

    def engine(self, batch, height, width, training=None):
        training = self.training if training is None else training
        key = (batch, height, width, bool(training))
        eng = self._engines.get(key)
        if eng is None:
            self._check_arena()
            eng = Engine(self, batch, height, width, bool(training), self._get_backend(), self._flat.device)
            eng.programs()
            # every engine owns a full activation set: keep the most recently used few (a short last batch, a
            # validation size), drop the oldest beyond that
            while len(self._engines) >= self.max_engines:
                self._engines.pop(next(iter(self._engines)))
            self._engines[key] = eng
        else:
            self._engines[key] = self._engines.pop(key)       # most recently used last
        return eng

    def refresh_weights(self):
        """bf16 shadow copies of the master weights: forward layout, padded stem rows, transposed copies."""
        be = self._get_backend()
        be.refresh_weights(self)

    # ------------------------------------------------------------------ passes
    def _run_forward(self, x, dropout_masks=None, generator=None, refresh=True):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('DenseASPP expects an N x 3 x H x W image batch, got %s' % (tuple(x.shape),))
        if not x.is_floating_point():
            raise TypeError('DenseASPP expects a floating point image batch')
        self._check_arena()
        if x.device != self._flat.device:
            raise RuntimeError('input is on %s but the model is on %s' % (x.device, self._flat.device))
        B, _, H, W = x.shape
        eng = self.engine(B, H, W)
        if refresh:
            self.refresh_weights()
        eng.set_input(x.detach().to(torch.float32))
        if eng.training:
            eng.draw_dropout(generator=generator, masks=dropout_masks)
        eng._fwd.run()
        return eng

    def forward(self, x):
        """Full-resolution logits N x n_class x H x W (fp32), differentiable w.r.t. the parameters."""
        if self.__dict__.get('_dp_master') is not None:
            return self._replica_forward(x)
        if torch.is_grad_enabled() and self.training:
            params = [p for _, p in self.named_parameters()]
            return _LogitsFn.apply(self, x, *params)
        eng = self._run_forward(x)
        eng.prog_up_fwd().run()
        sp = self.spec
        return eng.out_logits.view(eng.B, sp.n_class, eng.H, eng.W).clone()

    def forward_lowres(self, x, dropout_masks=None):
        """Logits at 1/8 resolution, N x n_class x H/8 x W/8 fp32 (no autograd)."""
        eng = self._run_forward(x, dropout_masks=dropout_masks)
        lg = eng.logits.view(eng.B, eng.H2, eng.W2, S.CLS_PAD)[..., :self.spec.n_class]
        return lg.permute(0, 3, 1, 2).contiguous()

    def predict(self, x):
        """Arg-max class per pixel, N x H x W int64 (fused upsample + argmax, eval-style)."""
        eng = self._run_forward(x)
        eng.prog_argmax().run()
        return eng.out_argmax.view(eng.B, eng.H, eng.W).clone()

    def train_step(self, x, labels, ignore_index=255, dropout_masks=None, generator=None, zero_grad=True,
                   refresh=True, reducer=None):
        """Fused forward + bilinear-x8/cross-entropy + backward.  Leaves gradients in the flat gradient
        arena (every Parameter's .grad is a view of it) and returns the mean loss as a 1-element tensor."""
        if not self.training:
            raise RuntimeError('train_step needs train() mode')
        if zero_grad:
            self._gflat_zero()
        eng = self._run_forward(x, dropout_masks=dropout_masks, generator=generator, refresh=refresh)
        if eng._loss_ignore != ignore_index:
            eng._fwd_loss = eng.backend.compile(eng.build_loss(ignore_index=ignore_index))
            eng._loss_ignore = ignore_index
        eng.set_labels(labels)
        eng._fwd_loss.run()
        if reducer is not None:
            reducer.run_backward()
        else:
            eng._bwd.run()
        g0 = self._first_param.grad
        if g0 is None or g0.data_ptr() != self._gflat.data_ptr() + 4 * self.arena.by_name[self._param_names[0]].off:
            self.bind_grads()
        return eng.loss

    def input_gradient(self, x):
        """Gradient of the loss of the most recent train_step (or of the backward pass just run through model(x)) with
        respect to the image batch x: N x 3 x H x W fp32.  x only selects the engine (its shape)."""
        B, _, H, W = x.shape
        eng = self._engines.get((B, H, W, True))
        if eng is None or eng._bwd is None:
            raise RuntimeError('input_gradient: no training step has run for a batch of shape %s' % (tuple(x.shape),))
        return eng.input_gradient().clone()

    def conv_macs(self, height, width):
        return self.spec.conv_macs(height, width)
