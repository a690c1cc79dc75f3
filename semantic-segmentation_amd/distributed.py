"""Data-parallel gradient reduction: one process per GPU, bucketed all-reduce over RCCL (xGMI),
overlapped with the backward pass.

The reference wraps the model in single-process nn.DataParallel (graph_seg_main.py:173), which
broadcasts all parameters and reduces all gradients through GPU 0 every step.  Here every rank
holds a full replica; because all gradients live in one flat arena in forward order and the
backward program finishes them from the end of the arena towards its start, a bucket is a
contiguous slice that becomes final at a known op index of the backward program.  The program is
replayed in segments; after each segment the finished slice is handed to an asynchronous
all-reduce, which RCCL runs on its own stream while the next segment computes.
"""
import torch
import torch.distributed as dist


# Compute units left to the collectives while buckets are in flight (of 256 on an MI355X; DSASPP_COMM_CUS overrides).
# The persistent kernels hold one 8-wave block with 200-250 registers per wave on every CU, i.e. the whole register file:
# a collective's workgroups cannot start on such a CU, and if they take CUs first, the persistent kernel's 256 blocks no
# longer fit in one round and it runs for up to twice its time.  16 CUs = two per XCD.  Measured on one MI355X, where
# a one-rank all-reduce launches no kernel, so only the cost shows (bench.py --force-reducer --comm-cus N, batch 32):
# 69.62 img/s with 0, 68.64 with 16 (-1.4 %), 67.31 with 32 (-3.3 %); without the reducer 69.58.
DEFAULT_COMM_CUS = 16


def broadcast_model(model, src=0, group=None):
    """Make every rank start from rank `src`'s replica: the flat fp32 parameter arena and every buffer (BatchNorm
    running statistics and batch counters) are broadcast, as DistributedDataParallel does when it wraps a module."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    model._check_arena()
    dist.broadcast(model._flat, src=src, group=group)
    # the buffers (three per BatchNorm, about 600 small tensors) travel as one flat tensor per dtype: one collective instead
    # of hundreds
    by_dtype = {}
    for _, b in model.named_buffers():
        by_dtype.setdefault(b.dtype, []).append(b)
    for bufs in by_dtype.values():
        flat = torch.cat([b.detach().reshape(-1) for b in bufs])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        with torch.no_grad():
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n
    # the bf16 / transposed shadow copies the kernels read were made from the old arena contents
    if getattr(model, '_backend', None) is not None:
        model.refresh_weights()


def all_reduce_mean(t, group=None):
    """Mean of a small tensor over the ranks (the loss value, for logging only)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


class GradReducer:
    def __init__(self, model, engine, bucket_mb=25.0, group=None, always=False, broadcast=True, comm_cus=None):
        self.model, self.engine, self.group = model, engine, group
        self.always = always             # issue the collectives even in a one-rank group (rehearsal of the N > 1 path)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Compute units the persistent (one block per CU) kernels of the backward pass leave free while collectives are
        # in flight, so that RCCL's reduction kernels have somewhere to run beside them (dsaspp_set_comm_cus).  Only
        # applied when collectives are actually issued; None = DEFAULT_COMM_CUS, 0 = none.
        if comm_cus is None:
            import os
            comm_cus = int(os.environ.get('DSASPP_COMM_CUS', DEFAULT_COMM_CUS))
        self.comm_cus = int(comm_cus)
        if broadcast:
            broadcast_model(model, 0, group)
        gflat = model._gflat
        n = gflat.numel()
        ops = engine._bwd.keepalive if hasattr(engine._bwd, 'keepalive') else engine._bwd.ops
        # last op index that writes each arena entry
        import bisect
        entries = sorted(model.arena.entries, key=lambda e: e.off)
        starts = [e.off for e in entries]
        last_write = {}
        for i, (kind, f) in enumerate(ops):
            for k, v in f.items():
                if hasattr(v, 't') and v.t is gflat:
                    # a descriptor may point into the middle of an entry (channel sub-range of a BatchNorm gradient)
                    j = bisect.bisect_right(starts, v.off) - 1
                    last_write[starts[j]] = i
        per_bucket = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.buckets = []          # (lo, hi, ready_op_index), from the end of the arena to its start
        hi = n
        ready = -1
        lo_cur = n
        for e in reversed(entries):
            ready = max(ready, last_write.get(e.off, -1))
            lo_cur = e.off
            if hi - lo_cur >= per_bucket:
                self.buckets.append((lo_cur, hi, ready))
                hi = lo_cur
                ready = -1
        if hi > 0:
            self.buckets.append((0, hi, max(ready, max(last_write.values()) if last_write else -1)))
        # ready indices must be non-decreasing along the bucket order for segment replay
        fixed = []
        run = -1
        for lo, hi_, r in self.buckets:
            run = max(run, r)
            fixed.append((lo, hi_, run))
        self.buckets = fixed
        self.n_ops = len(ops)

    def run_backward(self):
        """Replay the backward program in segments, launching one async all-reduce per finished bucket."""
        eng = self.engine
        gflat = self.model._gflat
        handles = []
        pos = 0
        comm = self.world > 1 or self.always
        reserve = comm and self.comm_cus > 0 and hasattr(eng.backend, 'lib')
        if reserve:
            from . import _lib
            before = _lib.get_comm_cus()
        try:
            for lo, hi, ready in self.buckets:
                end = ready + 1
                if end > pos:
                    eng._bwd.run(first=pos, last=end)
                    pos = end
                if comm:
                    handles.append(dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    if reserve and len(handles) == 1:
                        _lib.set_comm_cus(self.comm_cus)       # from the first collective on: launches read it one by one
            if pos < self.n_ops:
                eng._bwd.run(first=pos, last=self.n_ops)
        finally:
            if reserve:
                _lib.set_comm_cus(before)
        for h in handles:
            h.wait()
        # the collectives summed: the optimizer divides (FusedAdam.step reads this when no grad_scale is passed;
        # with another optimizer call average_grads() first)
        self.model._grad_scale = 1.0 / self.world
        return self.model._grad_scale

    def average_grads(self):
        """Divide the summed gradients in place, for optimizers that do not take a gradient scale (torch.optim.Adam)."""
        if self.world > 1:
            self.model._gflat.mul_(1.0 / self.world)
        self.model._grad_scale = 1.0
