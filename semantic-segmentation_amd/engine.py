"""Execution planner for the DenseASPP forward/backward path.

For one (batch, height, width, training) configuration the Engine owns every activation
buffer (feature slabs, bottleneck tensors, gradient slabs, statistics) and records the
forward and backward passes once as flat lists of kernel descriptors -- the same
descriptors include/dsaspp.h defines.  A backend turns a list into something runnable:
the product backend (HipBackend) copies the descriptors into a dsaspp_program inside
libdsaspp.so and replays it with one C call per pass; tests substitute a backend that
interprets the same descriptors on the CPU to check the orchestration without a GPU.

What replaces what (reference call sites):
  * torch.cat in every dense layer / ASPP branch -> convolutions write their g (or d1)
    output channels straight into the block's NHWC feature slab (graph_seg_main.py:237)
  * BatchNorm2d + ReLU in front of every conv -> per-consumer scale/shift tables applied in
    the conv prologue; batch statistics come from sums taken in the producer's epilogue
  * loss.backward() (graph_seg_main.py:239) -> per layer: weight gradient with the prologue
    recomputed, data gradient accumulated into an fp32 gradient slab.  BatchNorm backward
    of the shared slab is made single-pass: each consumer adds scale*dy and keeps
    (sum dy, sum dy*xhat); the mean terms are folded into two per-channel correction
    vectors that are applied when the producing layer reads its gradient back.
"""
import torch

from . import spec as S


class Ref:
    """A device tensor plus an element offset; what a descriptor pointer field holds."""
    __slots__ = ('t', 'off')

    def __init__(self, t, off=0):
        self.t, self.off = t, int(off)

    def __add__(self, k):
        return Ref(self.t, self.off + int(k))

    def ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()


# ops whose descriptor carries the activation storage type (include/dsaspp.h: `dtype`)
_DTYPE_OPS = frozenset(('conv', 'bn_act', 'avgpool_fwd', 'im2col', 'maxpool_fwd', 'stem3', 'dwconv', 'wgrad', 'grad_extract',
                        'bn_bwd_apply', 'avgpool_bwd', 'maxpool_bwd', 'col2im'))


class OpList(list):
    def __init__(self, dtype_code=0):
        super().__init__()
        self.dtype_code = dtype_code
        self.side = set()       # indices of ops that may run on the program's side stream (see dsaspp_program_set_side)
        self.join = set()       # indices of ops before which the caller's stream waits for the side stream

    def mark_side(self):
        self.side.add(len(self) - 1)

    def mark_join(self):
        self.join.add(len(self) - 1)

    def op(self, kind, **fields):
        if self.dtype_code and kind in _DTYPE_OPS:
            fields['dtype'] = self.dtype_code
        self.append((kind, fields))


def _r(t, off=0):
    return Ref(t, off)


class Slab:
    """NHWC feature slab of one dense block (or the ASPP stack) with its statistics."""

    def __init__(self, eng, M, C, H, W, train):
        dev = eng.device
        self.M, self.C, self.H, self.W = M, C, H, W
        self.x = torch.zeros(M * C, dtype=eng.backend.act_dtype, device=dev)
        self.mean = torch.zeros(C, dtype=torch.float32, device=dev)
        self.var = torch.ones(C, dtype=torch.float32, device=dev)
        self.rstd = torch.ones(C, dtype=torch.float32, device=dev)
        if train:
            self.g = torch.zeros(M * C, dtype=eng.grad_dtype, device=dev)
            self.corr = torch.zeros(2 * C, dtype=torch.float32, device=dev)   # corr1 | corr2


class Engine:
    def __init__(self, model, batch, height, width, training, backend, device):
        mult = getattr(model.spec, 'output_stride', 8)
        if height % mult or width % mult:
            raise ValueError('input height and width must be multiples of %d (got %dx%d)' % (mult, height, width))
        self.model = model
        self.spec = model.spec
        self.arena = model.arena
        self.B, self.H, self.W = batch, height, width
        self.training = training
        self.backend = backend
        self.device = device
        self.R = backend.stat_replicas
        # gradient slabs: bf16 halves the read-modify-write traffic of the slab-accumulating data
        # gradients (what mixed-precision autograd does for the cat() gradient); fp32 is the exact option
        self.grad_dtype = model.grad_slab_dtype if backend.act_dtype == torch.bfloat16 else torch.float32
        self.g_bf16 = int(self.grad_dtype == torch.bfloat16)
        sp = self.spec
        dev = device
        f32, bf16 = torch.float32, backend.act_dtype
        self.act_dtype = bf16
        self.dtype_code = 1 if backend.act_dtype == torch.float32 else 0
        if self.dtype_code and training and getattr(backend, 'forward_only_fp32', False) and (
                model.spec.kind == 'mobilenet' or getattr(model.spec, 'output_stride', 8) != 8):
            raise NotImplementedError('fp32 activation storage on the MI355X path trains the DenseNet trunks at output stride 8 '
                                      'only; train this configuration with the default bf16 storage')

        def zeros(n, dt=f32):
            return torch.zeros(int(n), dtype=dt, device=dev)

        self._zeros = zeros
        B = batch
        self.H0, self.W0 = height // 2, width // 2          # conv0 output
        self.H1, self.W1 = height // 4, width // 4          # block 1
        self.H2, self.W2 = height // 8, width // 8          # blocks 2-4, ASPP
        self.M0, self.M1, self.M2 = B * self.H0 * self.W0, B * self.H1 * self.W1, B * self.H2 * self.W2
        self.os16 = getattr(sp, 'output_stride', 8) == 16
        self.H3, self.W3 = height // 16, width // 16        # blocks 3-4 at output stride 16
        self.M3 = B * self.H3 * self.W3
        self.geo = {4: (self.M1, self.H1, self.W1), 8: (self.M2, self.H2, self.W2), 16: (self.M3, self.H3, self.W3)}

        # ---- activations
        self.mobilenet = sp.kind == 'mobilenet'
        self.fused_stem = False
        self.x_in = zeros(B * 3 * height * width)
        self.slabs = []
        self.aslab = Slab(self, self.M2, sp.C_aspp, self.H2, self.W2, training)
        self.logits = zeros(self.M2 * S.CLS_PAD)
        if self.mobilenet:
            self._alloc_mobilenet(zeros, bf16, training)
        else:
            # conv0 without a column matrix in memory (dsaspp_stem7_fwd / dsaspp_stem7_wgrad): bf16 storage, 64 or 96 channels
            self.fused_stem = (bool(getattr(model, 'fused_stem', True)) and self.dtype_code == 0 and sp.c_stem in (64, 96)
                               and hasattr(backend, 'has_op') and backend.has_op('stem7_fwd'))
            self.col = zeros(8 if self.fused_stem else self.M0 * S.STEM_KPAD, bf16)
            self.c0out = zeros(self.M0 * sp.c_stem, bf16)
            for blk in sp.blocks:
                M, H, W = self.geo[blk.res]
                self.slabs.append(Slab(self, M, blk.C, H, W, training))
            # un-pooled output of a transition that is followed by 2x2 average pooling
            self.t1tmp = zeros(max(self.geo[b.res][0] * b.cout for b in sp.blocks if b.pool_after), bf16)
            self.Mf = self.geo[sp.blocks[-1].res][0]           # pixels of the trunk output (norm5 input)
            self.t4out = zeros(self.Mf * sp.c_feat, bf16)
        # bottleneck tensors: kept per layer when training (needed by backward), shared otherwise
        self.y1 = {}
        shared = {}
        for blk in sp.blocks:
            M = self.geo[blk.res][0]
            for l in blk.layers:
                if training:
                    self.y1[l.prefix] = zeros(M * l.mid, bf16)
                else:
                    key = (M, l.mid)
                    if key not in shared:
                        shared[key] = zeros(M * l.mid, bf16)
                    self.y1[l.prefix] = shared[key]
        for a in sp.aspp:
            if training:
                self.y1[a.prefix] = zeros(self.M2 * a.d0, bf16)
            else:
                key = (self.M2, a.d0)
                if key not in shared:
                    shared[key] = zeros(self.M2 * a.d0, bf16)
                self.y1[a.prefix] = shared[key]

        # ---- per-BatchNorm tables: scale | shift | mean | var | rstd, C floats each
        self.bn = {}
        self.acc = {}
        self._pending_zero = []

        def bn_tables(name, C):
            self.bn[name] = zeros(5 * C)

        def accbuf(name, C):
            self.acc[name] = zeros(self.R * 2 * C)

        if self.mobilenet:
            bn_list = [(sp.stem_prefix + '.1', sp.c_stem)]
            for b in sp.ir:
                bn_list += [(b.prefix + '.conv.1', b.hid), (b.prefix + '.conv.4', b.hid), (b.prefix + '.conv.7', b.cout)]
            for name, c in bn_list:
                bn_tables(name, c)
                if training:
                    accbuf(name, c)            # forward statistics
                    accbuf(name + '.s', c)     # BatchNorm-backward sums
        else:
            bn_tables('features.norm0', sp.c_stem)
            accbuf('features.norm0', sp.c_stem)
            accbuf('pool0', sp.c_stem)
        for blk in sp.blocks:
            for l in blk.layers:
                bn_tables(l.prefix + '.norm1', l.cin)
                bn_tables(l.prefix + '.norm2', l.mid)
                accbuf(l.prefix + '.y1', l.mid)
                accbuf(l.prefix + '.out', l.growth)
                if training:
                    accbuf(l.prefix + '.s1', l.cin)
                    accbuf(l.prefix + '.s2', l.mid)
            bn_tables(blk.trans_prefix + '.norm', blk.C)
            accbuf(blk.trans_prefix + '.out', blk.cout)
            if training:
                accbuf(blk.trans_prefix + '.s', blk.C)
        if not self.mobilenet:
            bn_tables('features.norm5', sp.c_feat)
        accbuf('feat', sp.c_feat)
        if training:
            accbuf('features.norm5.s', sp.c_feat)
        for a in sp.aspp:
            if a.bn_start:
                bn_tables(a.prefix + '.norm1', a.cin)
                if training:
                    accbuf(a.prefix + '.s1', a.cin)
            bn_tables(a.prefix + '.norm2', a.d0)
            accbuf(a.prefix + '.y1', a.d0)
            accbuf(a.prefix + '.out', a.d1)
            if training:
                accbuf(a.prefix + '.s2', a.d0)

        # ---- loss / backward temporaries
        self.labels = torch.zeros(B * height * width, dtype=torch.int64, device=dev)
        self.accum = zeros(1024)      # DSASPP_CE_ACCUM_FLOATS: 32 (loss, count) pairs, one per 128-byte line
        self.loss = zeros(1)
        if training:
            self.dlogits = zeros(self.M2 * S.CLS_PAD)
            self.dlogits16 = zeros(self.M2 * S.CLS_PAD, bf16)
            self.argmax0 = torch.zeros(self.M1 * sp.c_stem, dtype=torch.uint8, device=dev)
            cmax_mid = max([l.mid for b in sp.blocks for l in b.layers] + [sp.d0])
            big = max(self.M0 * sp.c_stem, self.M2 * sp.c_feat, self.M1 * cmax_mid, self.M2 * cmax_mid,
                      *[self.geo[b.res][0] * b.cout for b in sp.blocks])
            if self.mobilenet:
                big = max(big, self.ir_n_e1, self.ir_n_io)
            self.tmp_a = zeros(big, bf16)
            # input of the BatchNorm backward (see bn_bwd_apply): bf16 like every other activation gradient, or fp32
            self.dr_f32 = int(getattr(model, 'bn_grad_dtype', torch.bfloat16) == torch.float32 or bf16 != torch.bfloat16)
            self.tmp_b = zeros(big) if self.dr_f32 else zeros(big, bf16)
            self.tmp_c = zeros(big, bf16)
            # bottleneck gradients of the layers whose slab data gradients are applied together
            self.dgrad_group = max(1, min(4, int(getattr(model, 'dgrad_group', 4))))
            self.wgrad_group = bool(getattr(model, 'wgrad_group', True))
            self.fuse_extract3 = bool(getattr(model, 'fuse_extract3', True))
            # the dense layers' 3x3 weight gradients on the program's side stream, next to the 3x3 data gradient of the
            # same layer (both are latency bound and read the same two tensors)
            self.side_wgrad3 = bool(getattr(model, 'side_wgrad3', False)) and not self.mobilenet
            self.ident_one = torch.ones(4096, dtype=f32, device=dev)
            self.ident_zero = zeros(4096)
            self.dy1_bufs = [self.tmp_c] + [zeros(max(self.M1, self.M2) * cmax_mid, bf16) for _ in range(self.dgrad_group - 1)]
            # Dense-layer weight gradients on the side stream: their gradient inputs must not be overwritten by the
            # next layers while they wait, so every dense layer gets private buffers for them (288 GB of HBM)
            self.side_wgrad = bool(getattr(model, 'side_wgrad', False)) and not self.mobilenet
            self.dz_l, self.dy1_l = {}, {}
            if self.side_wgrad:
                for blk in sp.blocks:
                    Mb = self.geo[blk.res][0]
                    for l in blk.layers:
                        self.dz_l[l.prefix] = zeros(Mb * l.growth, bf16)
                        self.dy1_l[l.prefix] = zeros(Mb * l.mid, bf16)
            self.coef = zeros(3 * max([sp.c_feat, cmax_mid, sp.c_stem] + [b.hid for b in getattr(sp, 'ir', ())]))
            self.dw0pad = zeros(sp.c_stem * max(S.STEM_KPAD, S.STEM3_KPAD))
            n_drop = 5 * sp.d1 + sp.C_aspp
            self.dropmask = torch.ones(B * n_drop, dtype=f32, device=dev)
            self.drop_ld = n_drop
        else:
            self.dropmask = None
        self.out_logits = None
        self.out_argmax = None
        self._fwd = None
        self._bwd = None
        self._fwd_loss = None
        self._up_fwd = None
        self._up_bwd = None
        self._argmax = None
        self._input_grad = None

    def _alloc_mobilenet(self, zeros, dt, training):
        """Trunk buffers of the MobileNetV2 variant.  Inference: two block in/out tensors (ping-pong) and the
        two expanded tensors of an inverted-residual block, each sized for the largest block.  Training keeps
        every block's expanded, depthwise, projected and output tensor for the backward pass."""
        sp = self.spec
        h, w = self.H0, self.W0
        self.stem_out = zeros(self.M0 * sp.c_stem, dt)
        n_io = n_e1 = n_e2 = 8
        self.ir_geom = []
        for b in sp.ir:
            ho, wo = (h - 1) // b.stride + 1, (w - 1) // b.stride + 1
            self.ir_geom.append((h, w, ho, wo))
            n_e1 = max(n_e1, self.B * h * w * b.hid)
            n_e2 = max(n_e2, self.B * ho * wo * b.hid)
            n_io = max(n_io, self.B * ho * wo * b.cout, self.B * h * w * b.cin)
            h, w = ho, wo
        if (h, w) != (self.H2, self.W2):
            raise ValueError('MobileNet trunk output %dx%d is not 1/8 of the input' % (h, w))
        self.ir_n_e1, self.ir_n_io = n_e1, n_io
        if not training:
            self.ir_io = [zeros(n_io, dt), zeros(n_io, dt)]
            self.ir_e1 = zeros(n_e1, dt)
            self.ir_e2 = zeros(n_e2, dt)
            self.ir_p = zeros(n_io, dt)
            return
        if sp.ir[-1].residual:
            raise NotImplementedError('MobileNet trunk whose last block has a skip connection')
        self.ir_keep = []
        for b, (h, w, ho, wo) in zip(sp.ir, self.ir_geom):
            last = b is sp.ir[-1]
            self.ir_keep.append(dict(e1=zeros(self.B * h * w * b.hid, dt), e2=zeros(self.B * ho * wo * b.hid, dt),
                                     p=zeros(self.B * ho * wo * b.cout, dt),
                                     out=None if last else zeros(self.B * ho * wo * b.cout, dt)))
        self.ir_g = [zeros(n_io, dt), zeros(n_io, dt)]         # block-output gradients (ping-pong)
        self.col3 = zeros(self.M0 * S.STEM3_KPAD, dt)          # 3x3/2 im2col rows for the stem weight gradient

    # ------------------------------------------------------------------ helpers
    def P(self, name):
        """Ref to a parameter in the fp32 arena."""
        return Ref(self.model._flat, self.arena.by_name[name].off)

    def PG(self, name):
        return Ref(self.model._gflat, self.arena.by_name[name].off)

    def W16(self, name):
        return Ref(self.model._w16, self.arena.by_name[name].off)

    def WT16(self, name):
        return Ref(self.model._wt16, self.arena.by_name[name].toff)

    def BUF(self, name):
        return Ref(self.model._buffers_flat[name], 0)

    def bnt(self, name, which, C, off=0):
        """Ref into a BatchNorm table: which in scale/shift/mean/var/rstd."""
        idx = ('scale', 'shift', 'mean', 'var', 'rstd').index(which)
        return Ref(self.bn[name], idx * C + off)

    def _bn_finalize(self, ops, name, C, slab=None, slab_off=0, acc=None, acc_c0=0, acc_cn=0, count=1.0,
                     affine=True, want_scale=True):
        """Record the forward BatchNorm bookkeeping for BatchNorm `name` over C channels."""
        m = self.model.bn_modules.get(name)
        use_running = not self.training
        if slab is not None:
            mean, var, rstd = _r(slab.mean, slab_off), _r(slab.var, slab_off), _r(slab.rstd, slab_off)
        else:
            mean, var, rstd = self.bnt(name, 'mean', C), self.bnt(name, 'var', C), self.bnt(name, 'rstd', C)
        f = dict(C=C, acc=acc, acc_c0=acc_c0, acc_cn=acc_cn, count=float(count),
                 mean=mean, var=var, rstd=rstd, use_running=int(use_running))
        if affine and m is not None:
            f.update(gamma=self.P(name + '.weight'), beta=self.P(name + '.bias'),
                     running_mean=self.BUF(name + '.running_mean'), running_var=self.BUF(name + '.running_var'),
                     num_batches_tracked=self.BUF(name + '.num_batches_tracked'),
                     momentum=float(m.momentum), eps=float(m.eps))
        else:
            f.update(gamma=None, beta=None, running_mean=None, running_var=None, num_batches_tracked=None,
                     momentum=0.0, eps=1e-5)
        if want_scale:
            f.update(scale=self.bnt(name, 'scale', C), shift=self.bnt(name, 'shift', C))
        else:
            f.update(scale=None, shift=None)
        ops.op('bn_finalize', **f)

    def _stem7(self, **kw):
        """Descriptor fields of the fused stem ops (forward and weight gradient share one struct)."""
        d = dict(n_img=self.B, H=self.H, W=self.W, x=_r(self.x_in), cout=self.spec.c_stem, w=None, ldw=0, y=None, ldy=0,
                 stat_acc=None, dr=None, lddr=0, dr_f32=0, yref=None, ldyref=0, coefA=None, coefB=None, coefC=None, dw=None,
                 lddw=0, ws=None, ws_floats=0)
        d.update(kw)
        return d

    # ------------------------------------------------------------------ forward
    def build_forward(self):
        sp, tr = self.spec, self.training
        ops = OpList(self.dtype_code)
        B = self.B
        conv = dict(in_scale=None, in_shift=None, in_relu=0, in_mask=None, in_mask_ld=0, bias=None,
                    out_mask=None, out_mask_ld=0, y=None, ldy=0, y_f32=0, stat_acc=None, epi_relu=0, ref=None,
                    ldref=0, e_scale=None, e_shift=None, e_mean=None, e_rstd=None, s_acc=None, g=None, ldg=0,
                    g_accumulate=0, g_bf16=self.g_bf16)
        self._conv_defaults = conv

        def C(**kw):
            d = dict(conv)
            d.update(kw)
            ops.op('conv', **d)

        if self.mobilenet:
            self._build_trunk_mobilenet(ops, C)
            if tr:
                self._feat_stats(ops)
            self._build_head(ops, C)
            return ops

        # ---- stem: im2col + GEMM, norm0 statistics, BN-ReLU-maxpool into slab 1
        cs = sp.c_stem
        if self.fused_stem:
            ops.op('stem7_fwd', **self._stem7(w=_r(self.model._w0_16), ldw=S.STEM_KPAD, y=_r(self.c0out), ldy=cs,
                                              stat_acc=_r(self.acc['features.norm0']) if tr else None))
        else:
            ops.op('im2col', n_img=B, H=self.H, W=self.W, x=_r(self.x_in), col=_r(self.col), ldcol=S.STEM_KPAD)
            C(n_img=B, H=self.H0, W=self.W0, cin=S.STEM_KPAD, cout=cs, taps=1, dil=1, x=_r(self.col), ldx=S.STEM_KPAD,
              w=_r(self.model._w0_16), y=_r(self.c0out), ldy=cs, stat_acc=_r(self.acc['features.norm0']) if tr else None)
        self._bn_finalize(ops, 'features.norm0', cs, acc=_r(self.acc['features.norm0']) if tr else None, acc_c0=0,
                          acc_cn=cs if tr else 0, count=self.M0)
        s1 = self.slabs[0]
        ops.op('maxpool_fwd', n_img=B, H=self.H0, W=self.W0, C=cs, x=_r(self.c0out), ldx=cs,
               scale=self.bnt('features.norm0', 'scale', cs), shift=self.bnt('features.norm0', 'shift', cs),
               y=_r(s1.x), ldy=s1.C, argmax=_r(self.argmax0) if tr else None,
               stat_acc=_r(self.acc['pool0']) if tr else None, dy=None, lddy=0, dr=None, lddr=0, dr_f32=0, mean=None, rstd=None,
               s_acc=None)
        pending = ('pool0', 0, cs)          # (acc name, first slab channel, count) still to be turned into mean/var

        # ---- dense blocks + transitions
        for bi, blk in enumerate(sp.blocks):
            slab = self.slabs[bi]
            M, H, W = slab.M, slab.H, slab.W
            for l in blk.layers:
                n1, n2 = l.prefix + '.norm1', l.prefix + '.norm2'
                self._bn_finalize(ops, n1, l.cin, slab=slab, acc=_r(self.acc[pending[0]]) if tr else None,
                                  acc_c0=pending[1], acc_cn=pending[2] if tr else 0, count=M)
                y1 = self.y1[l.prefix]
                C(n_img=B, H=H, W=W, cin=l.cin, cout=l.mid, taps=1, dil=1, x=_r(slab.x), ldx=slab.C,
                  w=self.W16(l.prefix + '.conv1.weight'), in_scale=self.bnt(n1, 'scale', l.cin),
                  in_shift=self.bnt(n1, 'shift', l.cin), in_relu=1, y=_r(y1), ldy=l.mid,
                  stat_acc=_r(self.acc[l.prefix + '.y1']) if tr else None)
                self._bn_finalize(ops, n2, l.mid, acc=_r(self.acc[l.prefix + '.y1']) if tr else None, acc_c0=0,
                                  acc_cn=l.mid if tr else 0, count=M)
                C(n_img=B, H=H, W=W, cin=l.mid, cout=l.growth, taps=9, dil=l.dil, x=_r(y1), ldx=l.mid,
                  w=self.W16(l.prefix + '.conv2.weight'), in_scale=self.bnt(n2, 'scale', l.mid),
                  in_shift=self.bnt(n2, 'shift', l.mid), in_relu=1, y=_r(slab.x, l.cin), ldy=slab.C,
                  stat_acc=_r(self.acc[l.prefix + '.out']) if tr else None)
                pending = (l.prefix + '.out', l.cin, l.growth)
            # transition
            nt = blk.trans_prefix + '.norm'
            self._bn_finalize(ops, nt, blk.C, slab=slab, acc=_r(self.acc[pending[0]]) if tr else None,
                              acc_c0=pending[1], acc_cn=pending[2] if tr else 0, count=M)
            tacc = _r(self.acc[blk.trans_prefix + '.out']) if tr else None
            common = dict(n_img=B, H=H, W=W, cin=blk.C, cout=blk.cout, taps=1, dil=1, x=_r(slab.x), ldx=slab.C,
                          w=self.W16(blk.trans_prefix + '.conv.weight'), in_scale=self.bnt(nt, 'scale', blk.C),
                          in_shift=self.bnt(nt, 'shift', blk.C), in_relu=1)
            if blk.pool_after:
                nxt = self.slabs[bi + 1]
                C(y=_r(self.t1tmp), ldy=blk.cout, **common)
                ops.op('avgpool_fwd', n_img=B, H=H, W=W, C=blk.cout, x=_r(self.t1tmp), ldx=blk.cout, y=_r(nxt.x),
                       ldy=nxt.C, stat_acc=tacc)
            elif bi + 1 < len(sp.blocks):
                nxt = self.slabs[bi + 1]
                C(y=_r(nxt.x), ldy=nxt.C, stat_acc=tacc, **common)
            else:
                C(y=_r(self.t4out), ldy=blk.cout, stat_acc=tacc, **common)
            pending = (blk.trans_prefix + '.out', 0, blk.cout)

        # ---- norm5 + ReLU into the ASPP slab; statistics of the result for the ASPP norm1 layers
        cf, A = sp.c_feat, self.aslab
        self._bn_finalize(ops, 'features.norm5', cf, acc=_r(self.acc[pending[0]]) if tr else None, acc_c0=0,
                          acc_cn=cf if tr else 0, count=self.Mf)
        if self.os16:
            # output stride 16: norm5 -> bilinear x2 -> (ReLU heading ASPP_3); the affine map commutes with the
            # interpolation, so one kernel reads the raw 1/16 map and writes ReLU(norm5) at 1/8 into the ASPP slab
            ops.op('up2_fwd', n_img=B, h=self.H3, w=self.W3, C=cf, x=_r(self.t4out), ldx=cf,
                   scale=self.bnt('features.norm5', 'scale', cf), shift=self.bnt('features.norm5', 'shift', cf), relu=1,
                   y=_r(A.x, sp.off_feat), ldy=A.C, stat_acc=_r(self.acc['feat']) if tr else None, dx=None, lddx=0, dx_f32=0,
                   mean=None, rstd=None, s_acc=None, dtype=self.dtype_code)
        else:
            ops.op('bn_act', M=self.M2, C=cf, x=_r(self.t4out), ldx=cf, scale=self.bnt('features.norm5', 'scale', cf),
                   shift=self.bnt('features.norm5', 'shift', cf), relu=1, y=_r(A.x, sp.off_feat), ldy=A.C,
                   stat_acc=_r(self.acc['feat']) if tr else None)
        if tr:
            self._feat_stats(ops)
        self._build_head(ops, C)
        return ops

    def _feat_stats(self, ops):
        """Batch statistics of the ASPP slab's feature channels, shared by every later ASPP norm1."""
        sp, A, cf = self.spec, self.aslab, self.spec.c_feat
        ops.op('bn_finalize', C=cf, acc=_r(self.acc['feat']), acc_c0=0, acc_cn=cf, count=float(self.M2),
               mean=_r(A.mean, sp.off_feat), var=_r(A.var, sp.off_feat), rstd=_r(A.rstd, sp.off_feat),
               gamma=None, beta=None, running_mean=None, running_var=None, num_batches_tracked=None,
               momentum=0.0, eps=1e-5, use_running=0, scale=None, shift=None)

    def _build_trunk_mobilenet(self, ops, C):
        """MobileNetV2 trunk: every BatchNorm is applied by its consumer -- in the prologue of the next
        convolution, or by the bn_act that closes a block (which also adds the skip connection).  In training
        the pointwise convolutions take the batch statistics of their outputs in their epilogues; the stem and
        the depthwise outputs get theirs from an in-place identity pass."""
        sp, B, tr = self.spec, self.B, self.training
        n0 = sp.stem_prefix + '.1'

        def acc(name):
            return _r(self.acc[name]) if tr else None

        def stats_pass(t, M, c_, name):
            ops.op('bn_act', M=M, C=c_, x=_r(t), ldx=c_, scale=None, shift=None, relu=0, y=_r(t), ldy=c_,
                   stat_acc=acc(name), res=None, ldres=0)

        def finalize(name, c_, count):
            self._bn_finalize(ops, name, c_, acc=acc(name), acc_c0=0, acc_cn=c_ if tr else 0, count=count)

        ops.op('stem3', n_img=B, H=self.H, W=self.W, x=_r(self.x_in), w=self.W16(sp.stem_prefix + '.0.weight'),
               cout=sp.c_stem, y=_r(self.stem_out), ldy=sp.c_stem)
        if tr:
            stats_pass(self.stem_out, self.M0, sp.c_stem, n0)
        finalize(n0, sp.c_stem, self.M0)
        cur, cur_c = self.stem_out, sp.c_stem
        cur_pro = dict(in_scale=self.bnt(n0, 'scale', sp.c_stem), in_shift=self.bnt(n0, 'shift', sp.c_stem), in_relu=2)
        A = self.aslab
        for i, b in enumerate(sp.ir):
            h, w, ho, wo = self.ir_geom[i]
            n1, n2, n3 = b.prefix + '.conv.1', b.prefix + '.conv.4', b.prefix + '.conv.7'
            if tr:
                keep = self.ir_keep[i]
                e1, e2, pj = keep['e1'], keep['e2'], keep['p']
            else:
                e1, e2, pj = self.ir_e1, self.ir_e2, self.ir_p
            C(n_img=B, H=h, W=w, cin=b.cin, cout=b.hid, taps=1, dil=1, x=_r(cur), ldx=cur_c,
              w=self.W16(b.prefix + '.conv.0.weight'), y=_r(e1), ldy=b.hid, stat_acc=acc(n1), **cur_pro)
            finalize(n1, b.hid, B * h * w)
            ops.op('dwconv', n_img=B, H=h, W=w, C=b.hid, stride=b.stride, dil=b.dil, x=_r(e1), ldx=b.hid,
                   in_scale=self.bnt(n1, 'scale', b.hid), in_shift=self.bnt(n1, 'shift', b.hid), in_relu=2,
                   w=self.W16(b.prefix + '.conv.3.weight'), y=_r(e2), ldy=b.hid)
            if tr:
                stats_pass(e2, B * ho * wo, b.hid, n2)
            finalize(n2, b.hid, B * ho * wo)
            C(n_img=B, H=ho, W=wo, cin=b.hid, cout=b.cout, taps=1, dil=1, x=_r(e2), ldx=b.hid,
              w=self.W16(b.prefix + '.conv.6.weight'), in_scale=self.bnt(n2, 'scale', b.hid),
              in_shift=self.bnt(n2, 'shift', b.hid), in_relu=2, y=_r(pj), ldy=b.cout, stat_acc=acc(n3))
            finalize(n3, b.cout, B * ho * wo)
            last = i == len(sp.ir) - 1
            if last:      # ReLU(trunk) goes straight into the ASPP slab (the in-place ReLU heading ASPP_3)
                dst, ldd = _r(A.x, sp.off_feat), A.C
            else:
                nxt = keep['out'] if tr else self.ir_io[i % 2]
                dst, ldd = _r(nxt), b.cout
            res = (_r(cur), cur_c) if b.residual else (None, 0)
            if b.residual and cur_pro:
                raise AssertionError('residual block fed by an unnormalised tensor')
            ops.op('bn_act', M=B * ho * wo, C=b.cout, x=_r(pj), ldx=b.cout, scale=self.bnt(n3, 'scale', b.cout),
                   shift=self.bnt(n3, 'shift', b.cout), relu=1 if last else 0, y=dst, ldy=ldd,
                   stat_acc=_r(self.acc['feat']) if (tr and last) else None, res=res[0], ldres=res[1])
            if not last:
                cur, cur_c, cur_pro = nxt, b.cout, {}

    def _build_head(self, ops, C):
        """ASPP stack + classifier on the ASPP slab whose feature channels hold ReLU(trunk output)."""
        sp, tr, B = self.spec, self.training, self.B
        A = self.aslab
        pending = None
        H, W, M = self.H2, self.W2, self.M2
        for i, a in enumerate(sp.aspp):
            y1 = self.y1[a.prefix]
            n2 = a.prefix + '.norm2'
            kw = dict(n_img=B, H=H, W=W, cin=a.cin, cout=a.d0, taps=1, dil=1, x=_r(A.x, a.off_in), ldx=A.C,
                      w=self.W16(a.prefix + '.conv1.weight'), bias=self.P(a.prefix + '.conv1.bias'), y=_r(y1),
                      ldy=a.d0, stat_acc=_r(self.acc[a.prefix + '.y1']) if tr else None)
            if a.bn_start:
                n1 = a.prefix + '.norm1'
                self._bn_finalize(ops, n1, a.cin, slab=A, slab_off=a.off_in,
                                  acc=_r(self.acc[pending[0]]) if tr else None, acc_c0=0,
                                  acc_cn=pending[2] if tr else 0, count=M)
                kw.update(in_scale=self.bnt(n1, 'scale', a.cin), in_shift=self.bnt(n1, 'shift', a.cin), in_relu=1)
            # ASPP_3 reads ReLU(norm5(x)) as stored: its own ReLU is the identity on that data
            C(**kw)
            self._bn_finalize(ops, n2, a.d0, acc=_r(self.acc[a.prefix + '.y1']) if tr else None, acc_c0=0,
                              acc_cn=a.d0 if tr else 0, count=M)
            last = i == len(sp.aspp) - 1
            C(n_img=B, H=H, W=W, cin=a.d0, cout=a.d1, taps=9, dil=a.rate, x=_r(y1), ldx=a.d0,
              w=self.W16(a.prefix + '.conv2.weight'), bias=self.P(a.prefix + '.conv2.bias'),
              in_scale=self.bnt(n2, 'scale', a.d0), in_shift=self.bnt(n2, 'shift', a.d0), in_relu=1,
              out_mask=_r(self.dropmask, i * sp.d1) if tr else None, out_mask_ld=self.drop_ld if tr else 0,
              y=_r(A.x, a.off_out), ldy=A.C, stat_acc=_r(self.acc[a.prefix + '.out']) if (tr and not last) else None)
            pending = (a.prefix + '.out', 0, a.d1)
        # ---- classifier (Dropout2d on its input in training) -> fp32 low-res logits
        C(n_img=B, H=H, W=W, cin=sp.C_aspp, cout=S.CLS_PAD, taps=1, dil=1, x=_r(A.x), ldx=A.C,
          w=self.W16('classification.1.weight'), bias=self.P('classification.1.bias'),
          in_mask=_r(self.dropmask, 5 * sp.d1) if tr else None, in_mask_ld=self.drop_ld if tr else 0,
          y=_r(self.logits), ldy=S.CLS_PAD, y_f32=1)

    # ------------------------------------------------------------------ loss (fused upsample + CE)
    def build_loss(self, ignore_index=255, grad_scale=1.0):
        sp = self.spec
        ops = OpList()
        ce = dict(n_img=self.B, h=self.H2, w=self.W2, n_class=sp.n_class, ldl=S.CLS_PAD, logits=_r(self.logits),
                  labels=_r(self.labels), ignore_index=int(ignore_index), dlogits=_r(self.dlogits),
                  accum=_r(self.accum), grad_scale=float(grad_scale), loss=_r(self.loss),
                  dlogits_bf16=None if self.dtype_code else _r(self.dlogits16), lddb=S.CLS_PAD,
                  dbias=self.PG('classification.1.bias'))
        ops.op('memset', src=None, dst=_r(self.dlogits), n=self.dlogits.numel() * 4, value=0, lds=0, ldd=0, rows=0,
               cols=0, dst_cols=0, accumulate=0)
        ops.op('memset', src=None, dst=_r(self.accum), n=4096, value=0, lds=0, ldd=0, rows=0, cols=0, dst_cols=0,
               accumulate=0)
        ops.op('upce', **ce)
        ops.op('ce_finalize', **ce)
        return ops

    # ------------------------------------------------------------------ backward
    def build_backward(self):
        """Backward pass from the low-resolution logit gradient (self.dlogits16, bf16 [M2][32])."""
        assert self.training
        sp = self.spec
        ops = OpList()
        B = self.B
        conv = self._conv_defaults
        A = self.aslab
        H, W, M = self.H2, self.W2, self.M2

        ops.dtype_code = self.dtype_code

        def C(**kw):
            d = dict(conv)
            d.update(kw)
            ops.op('conv', **d)

        def WG(**kw):
            d = dict(in_scale=None, in_shift=None, in_relu=0, in_mask=None, in_mask_ld=0, ksplit=0)
            d.update(kw)
            ops.op('wgrad', **d)

        def zero(t):
            ops.op('memset', src=None, dst=_r(t), n=t.numel() * t.element_size(), value=0, lds=0, ldd=0, rows=0,
                   cols=0, dst_cols=0, accumulate=0)

        def gradx(**kw):
            d = dict(hw=1, g_bf16=self.g_bf16, corr1=None, corr2=None, xs=None, ldxs=0, s_mean=None, s_rstd=None, mask=None, mask_ld=0,
                     relu=0, yref=None, ldyref=0, r_scale=None, r_shift=None, r_mean=None, r_rstd=None, s_acc=None,
                     colsum=None, dz_f32=0)
            d.update(kw)
            ops.op('grad_extract', **d)

        def slab_extract(slab, c0, cn, dz, **kw):
            """dz[M][cn] <- gradient of slab channels [c0, c0+cn) with the BatchNorm corrections applied."""
            gradx(M=slab.M, C=cn, g=_r(slab.g, c0), ldg=slab.C, corr1=_r(slab.corr, c0), corr2=_r(slab.corr, slab.C + c0),
                  xs=_r(slab.x, c0), ldxs=slab.C, s_mean=_r(slab.mean, c0), s_rstd=_r(slab.rstd, c0), dz=_r(dz), lddz=cn,
                  **kw)
            if self.side_wgrad3 and dz is self.tmp_a:
                ops.mark_join()         # a 3x3 weight gradient on the side stream may still be reading this buffer

        def bn_bwd(name, C_, acc, count, mode, slab=None, slab_off=0, own_stats=True):
            f = dict(C=C_, acc=_r(self.acc[acc]), count=float(count), scale=self.bnt(name, 'scale', C_),
                     dgamma=self.PG(name + '.weight'), dbeta=self.PG(name + '.bias'), mode=mode,
                     coefA=None, coefB=None, coefC=None, corr1=None, corr2=None, mean=None, rstd=None)
            if mode == 0:
                f.update(mean=self.bnt(name, 'mean', C_), rstd=self.bnt(name, 'rstd', C_),
                         coefA=_r(self.coef, 0), coefB=_r(self.coef, C_), coefC=_r(self.coef, 2 * C_))
            else:
                f.update(corr1=_r(slab.corr, slab_off), corr2=_r(slab.corr, slab.C + slab_off))
            ops.op('bn_bwd_finalize', **f)

        def bn_bwd_slice(name, Cfull, acc, count, slab, c0, cn):
            """BatchNorm backward bookkeeping of channels [c0, c0+cn) of a slab-input BatchNorm."""
            ops.op('bn_bwd_finalize', C=cn, acc=_r(self.acc[acc], c0), acc_ld=Cfull, count=float(count),
                   scale=self.bnt(name, 'scale', Cfull, c0), dgamma=self.PG(name + '.weight') + c0,
                   dbeta=self.PG(name + '.bias') + c0, mode=1, coefA=None, coefB=None, coefC=None,
                   corr1=_r(slab.corr, c0), corr2=_r(slab.corr, slab.C + c0), mean=None, rstd=None)

        def bn_apply(C_, M_, dr, y, dy, colsum=None):
            ops.op('bn_bwd_apply', M=M_, C=C_, dr=_r(dr), lddr=C_, dr_f32=self.dr_f32, y=_r(y), ldy=C_, coefA=_r(self.coef, 0),
                   coefB=_r(self.coef, C_), coefC=_r(self.coef, 2 * C_), dy=_r(dy), lddy=C_, colsum=colsum)

        dz, dr, dy1 = self.tmp_a, self.tmp_b, self.tmp_c
        # the low-resolution logit gradient the classifier's backward reads: the bf16 copy, or (fp32 storage) the fp32
        # tensor itself, which ce_finalize has normalised in place
        dlog = self.dlogits if self.dtype_code else self.dlogits16

        # ---- classifier
        zero(A.corr)
        mask_cls = _r(self.dropmask, 5 * sp.d1)
        WG(n_img=B, H=H, W=W, cin=sp.C_aspp, cout=S.CLS_PAD, taps=1, dil=1, x=_r(A.x), ldx=A.C, in_mask=mask_cls,
           in_mask_ld=self.drop_ld, dz=_r(dlog), lddz=S.CLS_PAD, dw=self.PG('classification.1.weight'))
        C(n_img=B, H=H, W=W, cin=S.CLS_PAD, cout=sp.C_aspp, taps=1, dil=1, x=_r(dlog), ldx=S.CLS_PAD,
          w=self.WT16('classification.1.weight'), out_mask=mask_cls, out_mask_ld=self.drop_ld, g=_r(A.g), ldg=A.C,
          g_accumulate=0)

        # ---- ASPP branches, last to first
        for i in reversed(range(len(sp.aspp))):
            a = sp.aspp[i]
            y1 = self.y1[a.prefix]
            n2 = a.prefix + '.norm2'
            slab_extract(A, a.off_out, a.d1, dz, hw=H * W, mask=_r(self.dropmask, i * sp.d1), mask_ld=self.drop_ld,
                         colsum=self.PG(a.prefix + '.conv2.bias'))
            WG(n_img=B, H=H, W=W, cin=a.d0, cout=a.d1, taps=9, dil=a.rate, x=_r(y1), ldx=a.d0,
               in_scale=self.bnt(n2, 'scale', a.d0), in_shift=self.bnt(n2, 'shift', a.d0), in_relu=1, dz=_r(dz),
               lddz=a.d1, dw=self.PG(a.prefix + '.conv2.weight'))
            C(n_img=B, H=H, W=W, cin=a.d1, cout=a.d0, taps=9, dil=a.rate, x=_r(dz), ldx=a.d1,
              w=self.WT16(a.prefix + '.conv2.weight'), y=_r(dr), ldy=a.d0, y_f32=self.dr_f32, epi_relu=1, ref=_r(y1), ldref=a.d0,
              e_scale=self.bnt(n2, 'scale', a.d0), e_shift=self.bnt(n2, 'shift', a.d0),
              e_mean=self.bnt(n2, 'mean', a.d0), e_rstd=self.bnt(n2, 'rstd', a.d0), s_acc=_r(self.acc[a.prefix + '.s2']))
            bn_bwd(n2, a.d0, a.prefix + '.s2', M, 0)
            bn_apply(a.d0, M, dr, y1, dy1, colsum=self.PG(a.prefix + '.conv1.bias'))
            if a.bn_start:
                n1 = a.prefix + '.norm1'
                WG(n_img=B, H=H, W=W, cin=a.cin, cout=a.d0, taps=1, dil=1, x=_r(A.x, a.off_in), ldx=A.C,
                   in_scale=self.bnt(n1, 'scale', a.cin), in_shift=self.bnt(n1, 'shift', a.cin), in_relu=1,
                   dz=_r(dy1), lddz=a.d0, dw=self.PG(a.prefix + '.conv1.weight'))
                C(n_img=B, H=H, W=W, cin=a.d0, cout=a.cin, taps=1, dil=1, x=_r(dy1), ldx=a.d0,
                  w=self.WT16(a.prefix + '.conv1.weight'), epi_relu=1, ref=_r(A.x, a.off_in), ldref=A.C,
                  e_scale=self.bnt(n1, 'scale', a.cin), e_shift=self.bnt(n1, 'shift', a.cin),
                  e_mean=_r(A.mean, a.off_in), e_rstd=_r(A.rstd, a.off_in), s_acc=_r(self.acc[a.prefix + '.s1']),
                  g=_r(A.g, a.off_in), ldg=A.C, g_accumulate=1)
                bn_bwd(n1, a.cin, a.prefix + '.s1', M, 1, slab=A, slab_off=a.off_in)
            else:
                WG(n_img=B, H=H, W=W, cin=a.cin, cout=a.d0, taps=1, dil=1, x=_r(A.x, a.off_in), ldx=A.C, dz=_r(dy1),
                   lddz=a.d0, dw=self.PG(a.prefix + '.conv1.weight'))
                # no BatchNorm in front of this block (its input is the ReLU output of norm5): the mask is that of the
                # stored value itself, written as the identity scale/shift so that the launch has the form of the
                # persistent slab data-gradient kernel
                C(n_img=B, H=H, W=W, cin=a.d0, cout=a.cin, taps=1, dil=1, x=_r(dy1), ldx=a.d0,
                  w=self.WT16(a.prefix + '.conv1.weight'), epi_relu=1, ref=_r(A.x, a.off_in), ldref=A.C,
                  e_scale=_r(self.ident_one), e_shift=_r(self.ident_zero), e_mean=_r(self.ident_zero),
                  e_rstd=_r(self.ident_one), g=_r(A.g, a.off_in), ldg=A.C, g_accumulate=1)

        if self.mobilenet:
            self._build_backward_mobilenet(ops, C, WG, gradx, slab_extract, bn_bwd, bn_apply, zero)
            return ops

        # ---- ReLU + norm5
        cf = sp.c_feat
        n5 = 'features.norm5'
        if self.os16:
            # ReLU mask from the stored (post-ReLU) slab values, then the adjoint of the x2 upsample with the
            # BatchNorm-backward sums of norm5 taken at 1/16 resolution
            slab_extract(A, sp.off_feat, cf, dy1, relu=1, yref=_r(A.x, sp.off_feat), ldyref=A.C)
            ops.op('up2_bwd', n_img=B, h=self.H3, w=self.W3, C=cf, x=_r(self.t4out), ldx=cf, scale=None, shift=None, relu=0,
                   y=_r(dy1), ldy=cf, stat_acc=None, dx=_r(dr), lddx=cf, dx_f32=self.dr_f32, mean=self.bnt(n5, 'mean', cf),
                   rstd=self.bnt(n5, 'rstd', cf), s_acc=_r(self.acc['features.norm5.s']), dtype=0)
        else:
            slab_extract(A, sp.off_feat, cf, dr, relu=1, yref=_r(self.t4out), ldyref=cf, r_scale=self.bnt(n5, 'scale', cf),
                         r_shift=self.bnt(n5, 'shift', cf), r_mean=self.bnt(n5, 'mean', cf), r_rstd=self.bnt(n5, 'rstd', cf),
                         s_acc=_r(self.acc['features.norm5.s']), dz_f32=self.dr_f32)
        bn_bwd(n5, cf, 'features.norm5.s', self.Mf, 0)
        bn_apply(cf, self.Mf, dr, self.t4out, dz)
        dz_cur = dz            # gradient w.r.t. the output of the last transition conv, [M2][cout]

        # ---- transitions and dense blocks, last to first
        for bi in reversed(range(len(sp.blocks))):
            blk = sp.blocks[bi]
            slab = self.slabs[bi]
            M, H, W = slab.M, slab.H, slab.W
            nt = blk.trans_prefix + '.norm'
            if bi + 1 < len(sp.blocks):
                nxt = self.slabs[bi + 1]
                if blk.pool_after:
                    slab_extract(nxt, 0, blk.cout, dy1)
                    ops.op('avgpool_bwd', n_img=B, H=H, W=W, C=blk.cout, x=_r(dz), ldx=blk.cout, y=_r(dy1),
                           ldy=blk.cout, stat_acc=None)
                    if self.side_wgrad3:
                        ops.mark_join()
                else:
                    slab_extract(nxt, 0, blk.cout, dz)
                dz_cur = dz
            zero(slab.corr)
            WG(n_img=B, H=H, W=W, cin=blk.C, cout=blk.cout, taps=1, dil=1, x=_r(slab.x), ldx=slab.C,
               in_scale=self.bnt(nt, 'scale', blk.C), in_shift=self.bnt(nt, 'shift', blk.C), in_relu=1, dz=_r(dz_cur),
               lddz=blk.cout, dw=self.PG(blk.trans_prefix + '.conv.weight'))
            C(n_img=B, H=H, W=W, cin=blk.cout, cout=blk.C, taps=1, dil=1, x=_r(dz_cur), ldx=blk.cout,
              w=self.WT16(blk.trans_prefix + '.conv.weight'), epi_relu=1, ref=_r(slab.x), ldref=slab.C,
              e_scale=self.bnt(nt, 'scale', blk.C), e_shift=self.bnt(nt, 'shift', blk.C), e_mean=_r(slab.mean),
              e_rstd=_r(slab.rstd), s_acc=_r(self.acc[blk.trans_prefix + '.s']), g=_r(slab.g), ldg=slab.C,
              g_accumulate=0)
            bn_bwd(nt, blk.C, blk.trans_prefix + '.s', M, 1, slab=slab, slab_off=0)
            # Layers are taken in groups, last layer first.  Within a group only the narrow channel
            # range the group itself produced is updated layer by layer; the wide slab prefix below it,
            # which every layer of the group feeds, is read and written once by a single multi-source
            # data-gradient launch after the group's bottleneck gradients exist.
            rev = list(reversed(blk.layers))
            gi = 0
            while gi < len(rev):
                grp = rev[gi:gi + self.dgrad_group]
                if any(l.mid != grp[0].mid for l in grp):
                    grp = grp[:1]
                gi += len(grp)
                c_lo = grp[-1].cin
                grp_dy1 = []
                # the group's bottleneck weight gradients in one launch (the slab prefix is streamed once for all of
                # them): needs the group's bottleneck gradients side by side, which the grouped data gradient needs too
                group_wgrad = self.wgrad_group and len(grp) > 1 and grp[0].mid <= 192
                for k, l in enumerate(grp):
                    n1, n2 = l.prefix + '.norm1', l.prefix + '.norm2'
                    y1 = self.y1[l.prefix]
                    side = self.side_wgrad
                    dy1k = self.dy1_l[l.prefix] if side else self.dy1_bufs[k]
                    dz_k = self.dz_l[l.prefix] if side else self.tmp_a
                    grp_dy1.append(dy1k)
                    # The read-out of the layer's gradient from the slab is the prologue of its 3x3 data gradient where the
                    # library has that form (bf16 slab and activations, the dense-layer shapes and dilations): the halo
                    # patch is read from the two slab slices directly and its own pixels leave as dz for the weight gradient.
                    fused = (self.fuse_extract3 and not self.dtype_code and self.g_bf16 and not self.dr_f32 and l.dil in (1, 2, 4)
                             and (l.growth, l.mid) in ((48, 192), (32, 128)))
                    conv3 = dict(n_img=B, H=H, W=W, cin=l.growth, cout=l.mid, taps=9, dil=l.dil,
                                 w=self.WT16(l.prefix + '.conv2.weight'), y=_r(dr), ldy=l.mid, y_f32=self.dr_f32, epi_relu=1, ref=_r(y1),
                                 ldref=l.mid, e_scale=self.bnt(n2, 'scale', l.mid), e_shift=self.bnt(n2, 'shift', l.mid),
                                 e_mean=self.bnt(n2, 'mean', l.mid), e_rstd=self.bnt(n2, 'rstd', l.mid),
                                 s_acc=_r(self.acc[l.prefix + '.s2']))
                    if fused:
                        C(x=_r(slab.g, l.cin), ldx=slab.C, gx_corr1=_r(slab.corr, l.cin), gx_corr2=_r(slab.corr, slab.C + l.cin),
                          gx_xs=_r(slab.x, l.cin), gx_ldxs=slab.C, gx_mean=_r(slab.mean, l.cin), gx_rstd=_r(slab.rstd, l.cin),
                          gx_dz=_r(dz_k), gx_lddz=l.growth, **conv3)
                        if self.side_wgrad3 and dz_k is self.tmp_a:
                            ops.mark_join()
                    else:
                        slab_extract(slab, l.cin, l.growth, dz_k)
                    WG(n_img=B, H=H, W=W, cin=l.mid, cout=l.growth, taps=9, dil=l.dil, x=_r(y1), ldx=l.mid,
                       in_scale=self.bnt(n2, 'scale', l.mid), in_shift=self.bnt(n2, 'shift', l.mid), in_relu=1, dz=_r(dz_k),
                       lddz=l.growth, dw=self.PG(l.prefix + '.conv2.weight'))
                    if side or self.side_wgrad3:
                        ops.mark_side()
                    if not fused:
                        C(x=_r(dz_k), ldx=l.growth, **conv3)
                    bn_bwd(n2, l.mid, l.prefix + '.s2', M, 0)
                    bn_apply(l.mid, M, dr, y1, dy1k)
                    if not group_wgrad:
                        WG(n_img=B, H=H, W=W, cin=l.cin, cout=l.mid, taps=1, dil=1, x=_r(slab.x), ldx=slab.C,
                           in_scale=self.bnt(n1, 'scale', l.cin), in_shift=self.bnt(n1, 'shift', l.cin), in_relu=1, dz=_r(dy1k),
                           lddz=l.mid, dw=self.PG(l.prefix + '.conv1.weight'))
                        if side:
                            ops.mark_side()
                    cn = l.cin - c_lo
                    if cn > 0:     # channels produced by the later-processed layers of this group
                        C(n_img=B, H=H, W=W, cin=l.mid, cout=cn, taps=1, dil=1, x=_r(dy1k), ldx=l.mid,
                          w=self.WT16(l.prefix + '.conv1.weight') + c_lo * l.mid, epi_relu=1, ref=_r(slab.x, c_lo),
                          ldref=slab.C, e_scale=self.bnt(n1, 'scale', l.cin, c_lo), e_shift=self.bnt(n1, 'shift', l.cin, c_lo),
                          e_mean=_r(slab.mean, c_lo), e_rstd=_r(slab.rstd, c_lo), s_acc=_r(self.acc[l.prefix + '.s1'], c_lo),
                          s_ld=l.cin, g=_r(slab.g, c_lo), ldg=slab.C, g_accumulate=1)
                        bn_bwd_slice(n1, l.cin, l.prefix + '.s1', M, slab, c_lo, cn)
                if group_wgrad:
                    wg = dict(n_src=len(grp))
                    for k, l in enumerate(grp):
                        n1 = l.prefix + '.norm1'
                        src = (_r(grp_dy1[k]), self.PG(l.prefix + '.conv1.weight'), self.bnt(n1, 'scale', l.cin),
                               self.bnt(n1, 'shift', l.cin), l.cin)
                        names = ('dz', 'dw', 'in_scale', 'in_shift', 'cin') if k == 0 else \
                            tuple(n + str(k) for n in ('dz', 'dw', 'sc', 'sh', 'cin'))
                        wg.update(zip(names, src))
                    WG(n_img=B, H=H, W=W, cout=grp[0].mid, taps=1, dil=1, x=_r(slab.x), ldx=slab.C, in_relu=1,
                       lddz=grp[0].mid, **wg)
                    if self.side_wgrad:
                        ops.mark_side()
                multi = dict(n_src=len(grp))
                for k, l in enumerate(grp):
                    n1 = l.prefix + '.norm1'
                    src = (_r(grp_dy1[k]), self.WT16(l.prefix + '.conv1.weight'), self.bnt(n1, 'scale', l.cin),
                           self.bnt(n1, 'shift', l.cin), _r(self.acc[l.prefix + '.s1']), l.cin)
                    names = ('x', 'w', 'e_scale', 'e_shift', 's_acc', 's_ld') if k == 0 else \
                        tuple(n + str(k) for n in ('x', 'w', 'es', 'eh', 'sa', 'sl'))
                    multi.update(zip(names, src))
                C(n_img=B, H=H, W=W, cin=grp[0].mid, cout=c_lo, taps=1, dil=1, ldx=grp[0].mid, epi_relu=1, ref=_r(slab.x),
                  ldref=slab.C, e_mean=_r(slab.mean), e_rstd=_r(slab.rstd), g=_r(slab.g), ldg=slab.C, g_accumulate=1, **multi)
                for l in grp:
                    bn_bwd_slice(l.prefix + '.norm1', l.cin, l.prefix + '.s1', M, slab, 0, c_lo)

        # ---- stem: maxpool + ReLU + norm0 backward, conv0 weight gradient
        cs = sp.c_stem
        s1 = self.slabs[0]
        n0 = 'features.norm0'
        slab_extract(s1, 0, cs, dz)
        ops.op('maxpool_bwd', n_img=B, H=self.H0, W=self.W0, C=cs, x=_r(self.c0out), ldx=cs,
               scale=self.bnt(n0, 'scale', cs), shift=self.bnt(n0, 'shift', cs), y=None, ldy=0, argmax=_r(self.argmax0),
               stat_acc=None, dy=_r(dz), lddy=cs, dr=_r(dr), lddr=cs, dr_f32=self.dr_f32, mean=self.bnt(n0, 'mean', cs),
               rstd=self.bnt(n0, 'rstd', cs), s_acc=_r(self.acc['features.norm0']))
        bn_bwd(n0, cs, 'features.norm0', self.M0, 0)
        if self.fused_stem:
            # BatchNorm-backward apply of norm0 and conv0's weight gradient in one kernel, column tiles rebuilt from the image
            if getattr(self, 'stem_ws', None) is None:
                n_ws = int(self.backend.lib.dsaspp_stem7_wgrad_workspace_floats(cs)) if hasattr(self.backend, 'lib') else 8
                self.stem_ws = self._zeros(n_ws)
            ops.op('stem7_wgrad', **self._stem7(dr=_r(dr), lddr=cs, dr_f32=self.dr_f32, yref=_r(self.c0out), ldyref=cs,
                                                coefA=_r(self.coef, 0), coefB=_r(self.coef, cs), coefC=_r(self.coef, 2 * cs),
                                                dw=self.PG('features.conv0.weight'), lddw=S.STEM_K, ws=_r(self.stem_ws),
                                                ws_floats=self.stem_ws.numel()))
            return ops
        bn_apply(cs, self.M0, dr, self.c0out, dy1)
        zero(self.dw0pad)
        WG(n_img=B, H=self.H0, W=self.W0, cin=S.STEM_KPAD, cout=cs, taps=1, dil=1, x=_r(self.col), ldx=S.STEM_KPAD,
           dz=_r(dy1), lddz=cs, dw=_r(self.dw0pad))
        ops.op('copy2d', src=_r(self.dw0pad), dst=self.PG('features.conv0.weight'), n=0, value=0, lds=S.STEM_KPAD,
               ldd=S.STEM_K, rows=cs, cols=S.STEM_K, dst_cols=S.STEM_K, accumulate=1)
        return ops

    def _build_backward_mobilenet(self, ops, C, WG, gradx, slab_extract, bn_bwd, bn_apply, zero):
        """Backward of the MobileNetV2 trunk, last block first.  Per block: BatchNorm backward of the projection
        (sums taken while the incoming gradient is read out), projection weight/data gradients (the data
        gradient masked by the ReLU6 of the depthwise BatchNorm and reduced to that BatchNorm's sums in the
        convolution epilogue), depthwise weight/data gradients (same fusion for the expansion BatchNorm),
        expansion weight/data gradients, skip-connection add."""
        sp, B, A = self.spec, self.B, self.aslab
        dz, dr, dy1 = self.tmp_a, self.tmp_b, self.tmp_c
        n0 = sp.stem_prefix + '.1'
        g_out = None                      # gradient w.r.t. the current block's output, bf16 [M][cout]
        for i in reversed(range(len(sp.ir))):
            b = sp.ir[i]
            h, w, ho, wo = self.ir_geom[i]
            Mi, Mo = B * h * w, B * ho * wo
            keep = self.ir_keep[i]
            e1, e2, pj = keep['e1'], keep['e2'], keep['p']
            n1, n2, n3 = b.prefix + '.conv.1', b.prefix + '.conv.4', b.prefix + '.conv.7'
            first, last = i == 0, i == len(sp.ir) - 1
            if first:
                x_in, c_in = self.stem_out, sp.c_stem
                pro = dict(in_scale=self.bnt(n0, 'scale', sp.c_stem), in_shift=self.bnt(n0, 'shift', sp.c_stem), in_relu=2)
            else:
                x_in, c_in, pro = self.ir_keep[i - 1]['out'], b.cin, {}
            # ---- BatchNorm of the projection
            stat3 = dict(r_mean=self.bnt(n3, 'mean', b.cout), r_rstd=self.bnt(n3, 'rstd', b.cout), s_acc=_r(self.acc[n3 + '.s']),
                         dz_f32=self.dr_f32)
            if last:      # from the ASPP slab, through the ReLU that heads the ASPP stack
                slab_extract(A, sp.off_feat, b.cout, dr, relu=1, yref=_r(pj), ldyref=b.cout,
                             r_scale=self.bnt(n3, 'scale', b.cout), r_shift=self.bnt(n3, 'shift', b.cout), **stat3)
            else:
                gradx(M=Mo, C=b.cout, g=_r(g_out), ldg=b.cout, g_bf16=1, yref=_r(pj), ldyref=b.cout, dz=_r(dr), lddz=b.cout, **stat3)
            bn_bwd(n3, b.cout, n3 + '.s', Mo, 0)
            bn_apply(b.cout, Mo, dr, pj, dz)
            # ---- projection (1x1, hid -> cout)
            WG(n_img=B, H=ho, W=wo, cin=b.hid, cout=b.cout, taps=1, dil=1, x=_r(e2), ldx=b.hid,
               in_scale=self.bnt(n2, 'scale', b.hid), in_shift=self.bnt(n2, 'shift', b.hid), in_relu=2, dz=_r(dz),
               lddz=b.cout, dw=self.PG(b.prefix + '.conv.6.weight'))
            C(n_img=B, H=ho, W=wo, cin=b.cout, cout=b.hid, taps=1, dil=1, x=_r(dz), ldx=b.cout,
              w=self.WT16(b.prefix + '.conv.6.weight'), y=_r(dr), ldy=b.hid, y_f32=self.dr_f32, epi_relu=2, ref=_r(e2),
              ldref=b.hid, e_scale=self.bnt(n2, 'scale', b.hid), e_shift=self.bnt(n2, 'shift', b.hid),
              e_mean=self.bnt(n2, 'mean', b.hid), e_rstd=self.bnt(n2, 'rstd', b.hid), s_acc=_r(self.acc[n2 + '.s']))
            bn_bwd(n2, b.hid, n2 + '.s', Mo, 0)
            bn_apply(b.hid, Mo, dr, e2, dy1)
            # ---- depthwise 3x3
            dwf = dict(n_img=B, H=h, W=w, C=b.hid, stride=b.stride, dil=b.dil, x=_r(e1), ldx=b.hid,
                       in_scale=self.bnt(n1, 'scale', b.hid), in_shift=self.bnt(n1, 'shift', b.hid), in_relu=2,
                       dy=_r(dy1), lddy=b.hid)
            ops.op('dw_wgrad', w=None, dx=None, lddx=0, dx_f32=0, e_mean=None, e_rstd=None, s_acc=None,
                   dw=self.PG(b.prefix + '.conv.3.weight'), **dwf)
            ops.op('dw_bwd_data', w=self.W16(b.prefix + '.conv.3.weight'), dx=_r(dr), lddx=b.hid, dx_f32=self.dr_f32,
                   e_mean=self.bnt(n1, 'mean', b.hid), e_rstd=self.bnt(n1, 'rstd', b.hid), s_acc=_r(self.acc[n1 + '.s']),
                   dw=None, **dwf)
            bn_bwd(n1, b.hid, n1 + '.s', Mi, 0)
            bn_apply(b.hid, Mi, dr, e1, dz)
            # ---- expansion (1x1, cin -> hid)
            WG(n_img=B, H=h, W=w, cin=c_in, cout=b.hid, taps=1, dil=1, x=_r(x_in), ldx=c_in, dz=_r(dz), lddz=b.hid,
               dw=self.PG(b.prefix + '.conv.0.weight'), **pro)
            dg = dict(n_img=B, H=h, W=w, cin=b.hid, cout=c_in, taps=1, dil=1, x=_r(dz), ldx=b.hid,
                      w=self.WT16(b.prefix + '.conv.0.weight'))
            if first:     # through the stem's ReLU6 + BatchNorm
                C(y=_r(dr), ldy=c_in, y_f32=self.dr_f32, epi_relu=2, ref=_r(self.stem_out), ldref=c_in,
                  e_scale=self.bnt(n0, 'scale', c_in), e_shift=self.bnt(n0, 'shift', c_in),
                  e_mean=self.bnt(n0, 'mean', c_in), e_rstd=self.bnt(n0, 'rstd', c_in), s_acc=_r(self.acc[n0 + '.s']), **dg)
            elif b.residual:
                g_in = self.ir_g[i % 2]
                C(y=_r(dy1), ldy=c_in, **dg)
                ops.op('bn_act', M=Mi, C=c_in, x=_r(dy1), ldx=c_in, scale=None, shift=None, relu=0, y=_r(g_in), ldy=c_in,
                       stat_acc=None, res=_r(g_out), ldres=c_in)
                g_out = g_in
            else:
                g_in = self.ir_g[i % 2]
                C(y=_r(g_in), ldy=c_in, **dg)
                g_out = g_in
        # ---- stem: BatchNorm backward, weight gradient as a GEMM over 3x3/2 im2col rows
        cs = sp.c_stem
        bn_bwd(n0, cs, n0 + '.s', self.M0, 0)
        bn_apply(cs, self.M0, dr, self.stem_out, dz)
        ops.op('stem3_col', n_img=B, H=self.H, W=self.W, x=_r(self.x_in), col=_r(self.col3), ldcol=S.STEM3_KPAD)
        zero(self.dw0pad)
        WG(n_img=B, H=self.H0, W=self.W0, cin=S.STEM3_KPAD, cout=cs, taps=1, dil=1, x=_r(self.col3), ldx=S.STEM3_KPAD,
           dz=_r(dz), lddz=cs, dw=_r(self.dw0pad))
        ops.op('copy2d', src=_r(self.dw0pad), dst=self.PG(sp.stem_prefix + '.0.weight'), n=0, value=0, lds=S.STEM3_KPAD,
               ldd=S.MOBILENET_STEM_K, rows=cs, cols=S.MOBILENET_STEM_K, dst_cols=S.MOBILENET_STEM_K, accumulate=1)

    # ------------------------------------------------------------------ full-resolution logits API
    def build_up_fwd(self):
        sp = self.spec
        if self.out_logits is None:
            self.out_logits = self._zeros(self.B * sp.n_class * self.H * self.W)
        ops = OpList()
        ops.op('up_fwd', n_img=self.B, h=self.H2, w=self.W2, n_class=sp.n_class, ldl=S.CLS_PAD, logits=_r(self.logits),
               out=_r(self.out_logits), argmax=None)
        return ops

    def build_argmax(self):
        sp = self.spec
        if self.out_argmax is None:
            self.out_argmax = torch.zeros(self.B * self.H * self.W, dtype=torch.int64, device=self.device)
        ops = OpList()
        ops.op('up_argmax', n_img=self.B, h=self.H2, w=self.W2, n_class=sp.n_class, ldl=S.CLS_PAD,
               logits=_r(self.logits), out=None, argmax=_r(self.out_argmax))
        return ops

    def build_up_bwd(self):
        """dlogits (fp32 low-res, then bf16 copy) from the full-resolution NCHW fp32 gradient in self.grad_full."""
        sp = self.spec
        self.grad_full = self._zeros(self.B * sp.n_class * self.H * self.W)
        self.accum_one = torch.zeros(1024, dtype=torch.float32, device=self.device)     # same layout as self.accum
        self.accum_one[1] = 1.0
        ops = OpList()
        ops.op('memset', src=None, dst=_r(self.dlogits), n=self.dlogits.numel() * 4, value=0, lds=0, ldd=0, rows=0,
               cols=0, dst_cols=0, accumulate=0)
        ops.op('up_bwd', n_img=self.B, h=self.H2, w=self.W2, n_class=sp.n_class, ldl=S.CLS_PAD, logits=_r(self.dlogits),
               out=_r(self.grad_full), argmax=None)
        # bf16 copy of the low-res gradient + classifier bias gradient: ce_finalize with the count pinned to 1
        ops.op('ce_finalize', n_img=self.B, h=self.H2, w=self.W2, n_class=sp.n_class, ldl=S.CLS_PAD, logits=None,
               labels=None, ignore_index=-1, dlogits=_r(self.dlogits), accum=_r(self.accum_one), grad_scale=1.0,
               loss=None, dlogits_bf16=None if self.dtype_code else _r(self.dlogits16), lddb=S.CLS_PAD,
               dbias=self.PG('classification.1.bias'))
        return ops

    # ------------------------------------------------------------------ gradient with respect to the input image
    def build_input_grad(self):
        """dx (fp32 NCHW) from the stem's output gradient, which the backward program leaves in place: the column gradient
        dcol = dy . W0 as a 1x1 convolution into the im2col buffer, then the adjoint of the im2col."""
        assert self.training
        sp, B = self.spec, self.B
        cs = sp.c_stem
        ops = OpList(self.dtype_code)
        if self.mobilenet:
            ks, kpad, dy, col = 3, S.STEM3_KPAD, self.tmp_a, self.col3
        elif self.fused_stem:
            # the fused weight gradient formed dy on the fly: form it here, into a column buffer of its own
            ks, kpad, dy = 7, S.STEM_KPAD, self.tmp_c
            col = self._zeros(self.M0 * kpad, self.act_dtype)
            ops.op('bn_bwd_apply', M=self.M0, C=cs, dr=_r(self.tmp_b), lddr=cs, dr_f32=self.dr_f32, y=_r(self.c0out), ldy=cs,
                   coefA=_r(self.coef, 0), coefB=_r(self.coef, cs), coefC=_r(self.coef, 2 * cs), dy=_r(dy), lddy=cs, colsum=None)
        else:
            ks, kpad, dy, col = 7, S.STEM_KPAD, self.tmp_c, self.col
        self.dx = self._zeros(B * 3 * self.H * self.W)
        self.w0t = self._zeros(kpad * cs, self.act_dtype)          # [k][cs]: W0 transposed, rows past ks*ks*3 zero
        d = dict(self._conv_defaults)
        d.update(n_img=B, H=self.H0, W=self.W0, cin=cs, cout=kpad, taps=1, dil=1, x=_r(dy), ldx=cs, w=_r(self.w0t), y=_r(col), ldy=kpad)
        ops.op('conv', **d)
        ops.op('col2im', n_img=B, H=self.H, W=self.W, ks=ks, dcol=_r(col), ldcol=kpad, dx=_r(self.dx), accumulate=0)
        return ops

    def input_gradient(self):
        """Run the input-gradient program (after a backward pass of this engine); returns the N x 3 x H x W fp32 tensor it
        owns."""
        if self._input_grad is None:
            self._input_grad = self.backend.compile(self.build_input_grad())
        sp = self.spec
        name = (sp.stem_prefix + '.0.weight') if self.mobilenet else 'features.conv0.weight'
        ks = 3 if self.mobilenet else 7
        w = self.arena.view(self.model._flat, name)                 # [cs, 3, ks, ks] view of the fp32 master weight
        k = ks * ks * 3
        wt = w.permute(2, 3, 1, 0).reshape(k, sp.c_stem)            # rows (ky, kx, ci): the im2col column order
        self.w0t.view(-1, sp.c_stem)[:k].copy_(wt)                  # rounds to the activation storage type
        self._input_grad.run()
        return self.dx.view(self.B, 3, self.H, self.W)

    # ------------------------------------------------------------------ running
    def programs(self):
        be = self.backend
        if self._fwd is None:
            self._fwd = be.compile(self.build_forward())
            if self.training:
                self._fwd_loss = be.compile(self.build_loss())
                self._loss_ignore = 255
                self._bwd = be.compile(self.build_backward())
        return self

    def prog_up_fwd(self):
        if self._up_fwd is None:
            self._up_fwd = self.backend.compile(self.build_up_fwd())
        return self._up_fwd

    def prog_up_bwd(self):
        if self._up_bwd is None:
            self._up_bwd = self.backend.compile(self.build_up_bwd())
        return self._up_bwd

    def prog_argmax(self):
        if self._argmax is None:
            self._argmax = self.backend.compile(self.build_argmax())
        return self._argmax

    def set_input(self, x):
        self.x_in.view(x.shape).copy_(x)

    def set_labels(self, labels):
        self.labels.view(labels.shape).copy_(labels)

    def draw_dropout(self, generator=None, masks=None):
        """Fill the per-(image, channel) Dropout2d scale tables (training only)."""
        sp = self.spec
        dm = self.dropmask.view(self.B, self.drop_ld)
        if masks is not None:
            for i, a in enumerate(sp.aspp):
                dm[:, i * sp.d1:(i + 1) * sp.d1] = masks[a.prefix].to(dm)
            dm[:, 5 * sp.d1:] = self._cls_mask_to_slab(masks['classification'].to(dm))
            return
        p0, p1 = sp.dropout0, sp.dropout1
        u = torch.rand(dm.shape, device=dm.device, generator=generator)
        n0 = 5 * sp.d1
        dm[:, :n0] = (u[:, :n0] >= p0).to(dm.dtype) / (1.0 - p0) if p0 > 0 else 1.0
        dm[:, n0:] = (u[:, n0:] >= p1).to(dm.dtype) / (1.0 - p1) if p1 > 0 else 1.0

    def _cls_mask_to_slab(self, m):
        # the classifier's input channel order IS the slab order ([ASPP_24 | ... | ASPP_3 | feature])
        return m
