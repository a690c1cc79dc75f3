"""CPU test (gloo, world_size 2): the bucketed, backward-overlapped gradient all-reduce.

Each rank runs the descriptor program through the CPU kernel model on its own batch; the reducer
replays the backward program in segments and all-reduces each finished slice of the flat gradient
arena.  Checks: (1) the segmented replay gives the same gradients as one uninterrupted replay,
(2) after reduction every rank holds the sum of both ranks' gradients, (3) the buckets tile the arena.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import kernel_model
        from e2e_common import TINY_CFG, make_batch, make_dropout_masks, make_pair
        from semantic_segmentation_amd.distributed import GradReducer
        kernel_model.set_act_dtype(torch.float32)
        ref, mod = make_pair(TINY_CFG)
        mod.set_backend(kernel_model.ModelBackend())
        mod.train()
        masks = make_dropout_masks(ref, 1, torch.Generator().manual_seed(5))
        grads = []
        for r in range(world):                       # every rank computes every rank's local gradient
            x, labels = make_batch(1, 64, 64, seed=10 + r)
            mod.train_step(x, labels, dropout_masks=masks)
            grads.append(mod._gflat.clone())
        x, labels = make_batch(1, 64, 64, seed=10 + rank)
        eng = mod.engine(1, 64, 64, True)
        red = GradReducer(mod, eng, bucket_mb=0.01)  # tiny buckets -> many segments
        assert len(red.buckets) > 3
        lo_prev = mod._gflat.numel()
        for lo, hi, ready in red.buckets:            # contiguous tiling from the end of the arena
            assert hi == lo_prev and lo < hi and 0 <= ready < red.n_ops
            lo_prev = lo
        assert lo_prev == 0
        assert [b[2] for b in red.buckets] == sorted(b[2] for b in red.buckets)
        loss = mod.train_step(x, labels, dropout_masks=masks, reducer=red)
        want = sum(grads)
        err = float((mod._gflat - want).abs().max() / want.abs().max())
        q.put((rank, err, float(loss)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, err, loss in res:
        assert err < 1e-5, (rank, err)
        assert loss == loss
