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


def _worker(rank, world, port, q, bucket_mb=0.01):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import kernel_model
        from e2e_common import TINY_CFG, make_batch, make_dropout_masks, make_pair
        from semantic_segmentation_amd.distributed import GradReducer, all_reduce_mean
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
        if rank > 0:                                 # replicas that differ until the reducer's broadcast from rank 0
            with torch.no_grad():
                mod._flat.add_(0.01 * rank)
                for n, b in mod.named_buffers():
                    if b.is_floating_point():
                        b.add_(0.5 * rank)
        red = GradReducer(mod, eng, bucket_mb=bucket_mb)  # tiny buckets -> many segments
        state = torch.cat([mod._flat] + [b.reshape(-1).float() for _, b in mod.named_buffers()])
        seen = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(seen, state)
        assert all(torch.equal(s, seen[0]) for s in seen), 'parameters / buffers were not broadcast from rank 0'
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
        assert mod._grad_scale == 1.0 / world
        mean_loss = float(all_reduce_mean(loss))
        red.average_grads()
        err2 = float((mod._gflat - want / world).abs().max() / want.abs().max())
        q.put((rank, max(err, err2), mean_loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,bucket_mb', [(2, 0.01), (4, 0.0137)])
def test_bucketed_allreduce(world, bucket_mb):
    """world 4 uses a bucket size that is no multiple of any parameter size, so bucket and segment boundaries fall
    unevenly; every rank must end with the same loss mean and the summed gradients."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bucket_mb)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    for rank, err, loss in res:
        assert err < 1e-5, (rank, err)
        assert loss == loss
    assert len({round(r[2], 6) for r in res}) == 1, 'the all-reduced loss differs between ranks'

