"""Allocation-free fused loss (dsaspp_upsample8_ce_loss_noalloc) against the two-call form with caller buffers
(dsaspp_upsample8_ce + dsaspp_ce_finalize), through the C ABI: two calls in a row, the second on a smaller input, so
that what the first call left in the built-in working memory would show."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from optools import Ref, close  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(n, h, w, seed, dev):
    gen = torch.Generator().manual_seed(seed)
    nc, ldl = 19, 32
    M = n * h * w
    logits = (torch.randn(M * ldl, generator=gen) * 2).to(dev)
    labels = torch.randint(0, nc, (n * 64 * h * w,), generator=gen)
    labels[torch.rand(labels.shape, generator=gen) < 0.1] = 255
    return dict(n_img=n, h=h, w=w, n_class=nc, ldl=ldl, logits=Ref(logits), labels=Ref(labels.to(dev)), ignore_index=255,
                grad_scale=0.5, lddb=ldl), M, ldl


def _outputs(M, ldl, dev):
    return dict(loss=Ref(torch.zeros(1, device=dev)), dlogits_bf16=Ref(torch.zeros(M * ldl, dtype=torch.bfloat16, device=dev)),
                dbias=Ref(torch.zeros(32, device=dev)))


def test_noalloc_loss_matches_two_call_form():
    from semantic_segmentation_amd import _lib
    from semantic_segmentation_amd.backend import HipBackend, fill_desc
    dev = 'cuda'
    be = HipBackend()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = 0
    for n, h, w, seed in ((3, 24, 40, 1), (1, 8, 16, 2)):       # the second call is smaller than the first
        f, M, ldl = _case(n, h, w, seed, dev)
        ref = _outputs(M, ldl, dev)
        ws = dict(dlogits=Ref(torch.zeros(M * ldl, device=dev)), accum=Ref(torch.zeros(1024, device=dev)))
        be.call('upce', **f, **ws, **ref)
        be.call('ce_finalize', **f, **ws, **ref)
        got = _outputs(M, ldl, dev)
        d = fill_desc(_lib.ce_desc, dict(f, dlogits=None, accum=None, **got))
        torch.cuda.synchronize()
        mem_before = torch.cuda.memory_allocated()
        free_before = torch.cuda.mem_get_info()[0]
        _lib.check(be.lib.dsaspp_upsample8_ce_loss_noalloc(ctypes.byref(d), stream))
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == mem_before       # nothing from torch's caching allocator ...
        calls += 1
        if calls > 1:
            # ... and nothing from the device itself (a hipMalloc inside the library would not pass through torch's
            # allocator).  Checked from the second call on: the first may load the code object / resolve the symbol.
            assert torch.cuda.mem_get_info()[0] == free_before
        # fp32 sums of atomically added terms: same tolerance as the op test of the two-call form
        close(ref['loss'].t.cpu(), got['loss'].t.cpu(), 1e-5, 'noalloc.loss')
        close(ref['dlogits_bf16'].t.cpu(), got['dlogits_bf16'].t.cpu(), 1e-2, 'noalloc.dlogits16')
        close(ref['dbias'].t.cpu(), got['dbias'].t.cpu(), 1e-4, 'noalloc.dbias')
        assert float(got['loss'].t) > 0


def test_noalloc_loss_rejects_oversize():
    from semantic_segmentation_amd import _lib
    from semantic_segmentation_amd.backend import HipBackend, fill_desc
    be = HipBackend()
    limit = _lib.DEFINES['DSASPP_NOALLOC_MAX_PIXELS']
    one = torch.zeros(8, device='cuda')
    lab = torch.zeros(8, dtype=torch.int64, device='cuda')
    # pointers are never dereferenced: the size check comes first
    d = fill_desc(_lib.ce_desc, dict(n_img=1, h=1, w=limit + 1, n_class=19, ldl=32, logits=Ref(one), labels=Ref(lab),
                                     ignore_index=255, grad_scale=1.0, loss=Ref(one), lddb=32))
    rc = be.lib.dsaspp_upsample8_ce_loss_noalloc(ctypes.byref(d), ctypes.c_void_p(0))
    assert rc != 0
    assert b'exceed' in be.lib.dsaspp_last_error()
