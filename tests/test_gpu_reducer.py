"""-m gpu: the bucketed gradient exchange on one MI355X (a one-rank RCCL group): segmented replay of the backward
program with the compute-unit reservation for the collectives, against the plain backward pass from the same forward
state."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_common import TINY_CFG, grad_report, make_batch, make_dropout_masks, make_pair  # noqa: E402


def test_comm_cus_knob_round_trip():
    import denseASPP  # noqa: F401
    from semantic_segmentation_amd import _lib as lib
    assert lib.get_comm_cus() == 0
    lib.set_comm_cus(24)
    assert lib.get_comm_cus() == 24
    lib.set_comm_cus(0)
    with pytest.raises(RuntimeError):
        lib.set_comm_cus(129)
    with pytest.raises(RuntimeError):
        lib.set_comm_cus(-1)
    assert lib.get_comm_cus() == 0


def test_segmented_backward_with_reserved_cus_matches_plain_backward(monkeypatch):
    import torch.distributed as dist
    from cfgs import DenseASPP121
    from semantic_segmentation_amd import _lib as lib
    from semantic_segmentation_amd.distributed import GradReducer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29571')
    started = not dist.is_initialized()
    if started:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        ref, mod = make_pair(DenseASPP121.Model_CFG, randomize_bn=False)
        B, H, W = 2, 128, 256
        x, labels = make_batch(B, H, W)
        masks = {k: v.cuda() for k, v in make_dropout_masks(ref, B, torch.Generator().manual_seed(5)).items()}
        mod = mod.cuda().train()
        loss_a = float(mod.train_step(x.cuda(), labels.cuda(), dropout_masks=masks))
        torch.cuda.synchronize()
        g_plain = {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters()}
        eng = mod.engine(B, H, W, True)
        red = GradReducer(mod, eng, bucket_mb=2.0, always=True, comm_cus=32)      # many small buckets
        assert len(red.buckets) > 4
        seen = []
        real = lib.set_comm_cus
        monkeypatch.setattr(lib, 'set_comm_cus', lambda n: (seen.append(int(n)), real(n))[1])
        # the backward again from the SAME forward state (the forward's statistics atomics make two forward passes differ
        # in the last bits, which the BatchNorm backward chain amplifies): loss kernel + segmented replay
        mod._gflat_zero()
        eng._fwd_loss.run()
        scale = red.run_backward()
        torch.cuda.synchronize()
        assert scale == 1.0 and seen == [32, 0], seen                           # reserved during the exchange, restored after
        assert lib.get_comm_cus() == 0
        g_red = {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters()}
        rel, cos, worst = grad_report(g_plain, g_red)
        print('segmented backward with 32 CUs reserved vs plain: rel-L2 %.2e cos %.7f (%s), loss %.5f' % (rel, cos, worst, loss_a))
        # the persistent kernels walk their tiles in another order on the smaller grid: summation order only, amplified
        # like the run-to-run differences of this network (test_gpu_e2e.py::test_fused_and_separate_gradient_readout_agree)
        head = sum(float((g_plain[k] - g_red[k]).pow(2).sum()) for k in g_plain if k.startswith(('classification', 'ASPP_')))
        head_n = sum(float(g_plain[k].pow(2).sum()) for k in g_plain if k.startswith(('classification', 'ASPP_')))
        assert (head / head_n) ** 0.5 < 1e-3
        assert rel < 0.1 and cos > 0.995, (rel, cos, worst)
    finally:
        if started:
            dist.destroy_process_group()
