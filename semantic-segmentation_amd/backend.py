"""The product backend: descriptors -> dsaspp_program inside libdsaspp.so (HIP, gfx950)."""
import ctypes

import torch

from . import _lib


class HipProgram:
    def __init__(self, lib, handle, keepalive):
        self.lib, self.handle, self.keepalive = lib, handle, keepalive

    def run(self, stream=None, first=0, last=-1):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.dsaspp_program_run(self.handle, first, last, ctypes.c_void_p(stream)))

    def __len__(self):
        return self.lib.dsaspp_program_size(self.handle)

    def set_tag(self, op_index, tag):
        _lib.check(self.lib.dsaspp_program_set_tag(self.handle, op_index, tag))

    def read_timing(self, n_tags):
        ms = (ctypes.c_double * n_tags)()
        cnt = (ctypes.c_int64 * n_tags)()
        _lib.check(self.lib.dsaspp_program_read_timing(self.handle, n_tags, ms, cnt))
        return list(ms), list(cnt)

    def __del__(self):
        try:
            self.lib.dsaspp_program_destroy(self.handle)
        except Exception:
            pass


def fill_desc(struct, fields):
    d = struct()
    known = {n for n, _ in struct._fields_}
    for k, v in fields.items():
        if k not in known:
            raise KeyError('%s has no field %r' % (struct.__name__, k))
        if v is None:
            continue
        if hasattr(v, 'ptr'):
            v = v.ptr()
        setattr(d, k, v)
    return d


class HipBackend:
    """Runs descriptor lists through libdsaspp.so.  There is no other compute backend in the package."""
    stat_replicas = _lib.STAT_REPLICAS
    act_dtype = torch.bfloat16
    forward_only_fp32 = True      # fp32 storage: kernels read the fp32 master weights; training only for DenseNet trunks at OS 8

    def __init__(self, act_dtype=torch.bfloat16):
        if act_dtype not in (torch.bfloat16, torch.float32):
            raise TypeError('activation storage must be torch.bfloat16 or torch.float32')
        self.lib = _lib.load()
        self.act_dtype = act_dtype

    @staticmethod
    def has_op(kind):
        return kind in _lib.OPS

    def compile(self, oplist):
        h = self.lib.dsaspp_program_create()
        if not h:
            raise _lib.DsasppError('dsaspp_program_create failed')
        prog = HipProgram(self.lib, h, oplist)
        for kind, fields in oplist:
            code, struct, _ = _lib.OPS[kind]
            d = fill_desc(struct, fields)
            _lib.check(self.lib.dsaspp_program_add(h, code, ctypes.byref(d), ctypes.sizeof(d)))
        for i in sorted(getattr(oplist, 'side', ())):
            _lib.check(self.lib.dsaspp_program_set_side(h, i, 1))
        for i in sorted(getattr(oplist, 'join', ())):
            _lib.check(self.lib.dsaspp_program_set_side(h, i, 2))
        return prog

    def call(self, kind, stream=None, **fields):
        """Launch one kernel directly (op-level tests)."""
        code, struct, entry = _lib.OPS[kind]
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        d = fill_desc(struct, fields)
        _lib.check(getattr(self.lib, entry)(ctypes.byref(d), ctypes.c_void_p(stream)))

    # ------------------------------------------------------------------ weight shadows
    @staticmethod
    def _wt_table(model):
        """Device table of (source offset, destination offset, shape) of every convolution weight, built once."""
        if model._wt_table is None:
            ents = [e for e in model.arena.entries if e.kind == 'conv']
            tab = (_lib.wt_entry * len(ents))()
            mx = 0
            for i, e in enumerate(ents):
                tab[i].src_off, tab[i].dst_off = e.off, e.toff
                tab[i].cout, tab[i].cin, tab[i].taps = e.rows, e.cin, e.taps
                mx = max(mx, e.numel)
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).clone()
            model._wt_table = (raw.to(model._flat.device), len(ents), mx)
        return model._wt_table

    def refresh_weights(self, model, cast=True):
        """fp32 arena -> bf16 shadow (same offsets), padded stem rows, transposed copies for the dgrad convs."""
        from . import spec as S
        lib = self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ar = model.arena
        if self.act_dtype == torch.float32:
            # fp32 form: the kernels read the fp32 master arena itself (model._w16 aliases it); only the
            # row-padded copy of the 7x7 stem weight has to be refreshed
            e0 = ar.by_name.get('features.conv0.weight')
            if e0 is not None:
                _lib.check(lib.dsaspp_copy2d_f32(model._flat.data_ptr() + 4 * e0.off, S.STEM_K, model._w0_16.data_ptr(),
                                                 S.STEM_KPAD, e0.rows, S.STEM_K, S.STEM_KPAD, 0, stream))
            # transposed fp32 copies for the data-gradient convolutions of the fp32 training form
            tab_dev, n, mx = self._wt_table(model)
            _lib.check(lib.dsaspp_weight_transpose_f32(model._flat.data_ptr(), model._wt16.data_ptr(), tab_dev.data_ptr(), n, mx,
                                                       stream))
            return
        if cast:
            _lib.check(lib.dsaspp_cast_bf16(model._flat.data_ptr(), model._w16.data_ptr(), ar.size, stream))
        e0 = ar.by_name.get('features.conv0.weight')
        if e0 is not None:      # 7x7 stem of the DenseNet trunk: rows padded 147 -> 160 for the im2col GEMM
            _lib.check(lib.dsaspp_copy2d_f32_bf16(model._flat.data_ptr() + 4 * e0.off, S.STEM_K, model._w0_16.data_ptr(),
                                                  S.STEM_KPAD, e0.rows, S.STEM_K, S.STEM_KPAD, stream))
        tab_dev, n, mx = self._wt_table(model)
        _lib.check(lib.dsaspp_weight_transpose(model._w16.data_ptr(), model._wt16.data_ptr(), tab_dev.data_ptr(), n, mx,
                                               stream))
