"""CPU tests (no GPU): the Engine's orchestration run through the CPU kernel model against the oracle.

With fp32 activation storage the descriptor program must reproduce the oracle's loss, every parameter
gradient and the BatchNorm running statistics to fp32 accuracy -- this pins the slab layout, the
"concatenate in front" ASPP ordering and the single-pass BatchNorm-backward correction scheme.
With bf16 storage (what the HIP kernels do) the logits must stay within the 2e-2 budget.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

import kernel_model
from e2e_common import TINY_CFG, grad_report, make_batch, make_dropout_masks, make_pair, oracle_train


@pytest.fixture
def fp32_model():
    kernel_model.set_act_dtype(torch.float32)
    yield
    kernel_model.set_act_dtype(torch.bfloat16)


def test_eval_forward_fp32(fp32_model):
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    x, _ = make_batch(2, 64, 96)
    ref.eval(); mod.eval()
    with torch.no_grad():
        yr, ym = ref(x), mod(x)
    assert ym.shape == yr.shape == (2, 19, 64, 96)
    assert float((yr - ym).abs().max() / yr.abs().max()) < 1e-4      # fp32 tolerance of the north star
    am = mod.predict(x)
    assert float((am == yr.argmax(1)).float().mean()) > 0.999


def test_train_step_fp32(fp32_model):
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(2, 64, 96)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod.train()
    loss_m = mod.train_step(x, labels, dropout_masks=masks)
    assert abs(float(loss_r) - float(loss_m)) < 1e-5
    rel, cos, worst = grad_report(g_ref, {n: p.grad for n, p in mod.named_parameters()})
    assert rel < 1e-3 and cos > 0.99999, (rel, cos, worst)
    br = dict(ref.named_buffers())
    for n, b in mod.named_buffers():
        assert float((br[n].float() - b.float()).abs().max()) < 1e-4, n


@pytest.mark.parametrize('group', [1, 3, 4])
def test_grouped_dense_backward_fp32(fp32_model, group):
    """Blocks of 6, 4, 2 and 3 layers: full groups, remainder groups and single-layer groups of the
    grouped slab data gradient all reproduce the oracle's gradients."""
    cfg = dict(TINY_CFG)
    cfg['block_config'] = (6, 4, 2, 3)
    ref, mod = make_pair(cfg)
    mod.dgrad_group = group
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(1, 32, 48)
    masks = make_dropout_masks(ref, 1, torch.Generator().manual_seed(7))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod.train()
    loss_m = mod.train_step(x, labels, dropout_masks=masks)
    assert abs(float(loss_r) - float(loss_m)) < 1e-5
    rel, cos, worst = grad_report(g_ref, {n: p.grad for n, p in mod.named_parameters()})
    assert rel < 1e-3 and cos > 0.99999, (rel, cos, worst)


def test_autograd_path_matches_fused_path(fp32_model):
    """model(x) -> CrossEntropyLoss -> backward (the reference's loop, graph_seg_main.py:237-239)."""
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(1, 64, 64)
    mod.train()
    masks = make_dropout_masks(ref, 1, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    eng = mod.engine(1, 64, 64, True)
    orig = eng.draw_dropout
    eng.draw_dropout = lambda generator=None, masks=None: orig(masks=masks if masks is not None else _masks)
    _masks = masks
    out = mod(x)
    assert out.requires_grad
    loss = torch.nn.CrossEntropyLoss(ignore_index=255)(out, labels)
    loss.backward()
    assert abs(float(loss) - float(loss_r)) < 1e-5
    rel, cos, worst = grad_report(g_ref, {n: p.grad for n, p in mod.named_parameters()})
    assert rel < 1e-3, (rel, cos, worst)


def test_bf16_storage_budget():
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    x, _ = make_batch(2, 64, 96)
    ref.eval(); mod.eval()
    with torch.no_grad():
        yr, ym = ref(x), mod(x)
    assert float((yr - ym).abs().max() / yr.abs().max()) < 2e-2


def test_state_dict_compat():
    ref, mod = make_pair(TINY_CFG)
    sd_r, sd_m = ref.state_dict(), mod.state_dict()
    assert list(sd_r.keys()) == list(sd_m.keys())
    for k in sd_r:
        assert sd_r[k].shape == sd_m[k].shape and torch.equal(sd_r[k], sd_m[k]), k
    # DataParallel-style and legacy dotted keys load too
    legacy = {}
    for k, v in sd_r.items():
        k2 = 'module.' + k.replace('.norm1.', '.norm.1.').replace('.conv2.', '.conv.2.')
        legacy[k2] = v
    import denseASPP
    m2 = denseASPP.DenseASPP(TINY_CFG)
    m2.load_state_dict(legacy)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd_r[k]), k
    wrapped = torch.nn.DataParallel(mod)
    assert all(k.startswith('module.') for k in wrapped.state_dict())
    wrapped.load_state_dict(wrapped.state_dict())


def test_ctor_forms():
    import denseASPP
    a = denseASPP.DenseASPP(TINY_CFG, 19, 8)
    b = denseASPP.DenseASPP(object(), model_cfg=TINY_CFG)
    c = denseASPP.DenseASPP(TINY_CFG, n_class=7)
    assert a.spec.n_class == b.spec.n_class == 19 and c.spec.n_class == 7
    assert denseASPP.DenseASPP(TINY_CFG, output_stride=16).spec.output_stride == 16
    with pytest.raises(ValueError):
        denseASPP.DenseASPP(TINY_CFG, output_stride=32)
    with pytest.raises(TypeError):
        denseASPP.DenseASPP()


def test_output_stride_16_fp32(fp32_model):
    """DenseASPP(model_cfg, n_class, output_stride=16): transition2 pools, blocks 3-4 run at 1/16 with halved
    dilations and norm5 is followed by a bilinear x2 upsample; forward, loss, every gradient and the running
    statistics reproduce the oracle."""
    ref, mod = make_pair(TINY_CFG, output_stride=16)
    mod.set_backend(kernel_model.ModelBackend())
    assert list(ref.state_dict().keys()) == list(mod.state_dict().keys())
    x, labels = make_batch(2, 64, 96)
    ref.eval(); mod.eval()
    with torch.no_grad():
        yr, ym = ref(x), mod(x)
    assert ym.shape == yr.shape == (2, 19, 64, 96)
    assert float((yr - ym).abs().max() / yr.abs().max()) < 1e-4
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    loss_r, g_ref = oracle_train(ref, x, labels, masks)
    mod.train()
    loss_m = mod.train_step(x, labels, dropout_masks=masks)
    assert abs(float(loss_r) - float(loss_m)) < 1e-5
    rel, cos, worst = grad_report(g_ref, {n: p.grad for n, p in mod.named_parameters()})
    assert rel < 1e-3 and cos > 0.99999, (rel, cos, worst)
    br = dict(ref.named_buffers())
    for n, b in mod.named_buffers():
        assert float((br[n].float() - b.float()).abs().max()) < 1e-4, n
    with pytest.raises(ValueError):
        mod.eval()(torch.zeros(1, 3, 72, 96))          # 72 is a multiple of 8 but not of 16


def test_legacy_keys_through_dataparallel_wrapper():
    """Upstream checkpoints name the children 'norm.1' / 'conv.2' and are loaded INTO the nn.DataParallel wrapper
    (graph_seg_test.py:131-135): the key hook then runs with prefix 'module.' and must still rewrite ASPP keys."""
    import denseASPP
    ref, mod = make_pair(TINY_CFG)
    sd = ref.state_dict()
    legacy = {}
    for k, v in sd.items():
        k2 = k
        for a, b in (('.norm1.', '.norm.1.'), ('.norm2.', '.norm.2.'), ('.conv1.', '.conv.1.'), ('.conv2.', '.conv.2.')):
            k2 = k2.replace(a, b)
        legacy['module.' + k2] = v
    assert any(k.startswith('module.ASPP_6.conv.1.') for k in legacy)
    wrapped = torch.nn.DataParallel(denseASPP.DenseASPP(TINY_CFG))
    wrapped.load_state_dict(legacy)                  # strict
    for k, v in wrapped.module.state_dict().items():
        assert torch.equal(v, sd[k]), k


def _fake_replicate(network, n):
    """What torch.nn.parallel.replicate does, with autograd-connected CPU copies instead of a CUDA broadcast."""
    params = list(network.parameters())
    pidx = {p: i for i, p in enumerate(params)}
    pcopies = [[p * 1.0 for p in params] for _ in range(n)]
    bufs = list(network.buffers())
    bidx = {b: i for i, b in enumerate(bufs)}
    bcopies = [[b.clone() for b in bufs] for _ in range(n)]
    modules = list(network.modules())
    midx = {m: i for i, m in enumerate(modules)}
    copies = [[m._replicate_for_data_parallel() for m in modules] for _ in range(n)]
    for i, m in enumerate(modules):
        for key, child in m._modules.items():
            for j in range(n):
                setattr(copies[j][i], key, copies[j][midx[child]])
        for key, p in m._parameters.items():
            for j in range(n):
                if p is None:
                    copies[j][i]._parameters[key] = None
                else:
                    setattr(copies[j][i], key, pcopies[j][pidx[p]])
        for key, b in m._buffers.items():
            for j in range(n):
                if b is None:
                    copies[j][i]._buffers[key] = None
                else:
                    setattr(copies[j][i], key, bcopies[j][bidx[b]])
    return [copies[j][0] for j in range(n)]


def test_dataparallel_replicas(fp32_model):
    """nn.DataParallel(model) without device_ids (graph_seg_main.py:173) replicates the module per device and step.
    The replica mechanics are exercised here on the CPU: two replicas made the way torch's replicate() makes them,
    one served by the master model and one by a separate worker copy, each on its half of the batch; the gradients
    that flow back to the master parameters must equal those of the two halves run one after the other."""
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    mod.train()
    x, labels = make_batch(2, 64, 96)
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    want = None
    for i in range(2):
        for p in mod.parameters():
            p.grad = None
        torch.manual_seed(3 + i)                     # the Dropout2d masks are drawn inside the forward call
        crit(mod(x[i:i + 1]), labels[i:i + 1]).backward()
        g = torch.cat([p.grad.reshape(-1) for p in mod.parameters()])
        want = g if want is None else want + g
    for p in mod.parameters():
        p.grad = None
    reps = _fake_replicate(mod, 2)
    assert all(r._dp_master is mod for r in reps)
    mod._dp_force_worker = True                      # the second replica runs on a worker copy, as on another device
    with pytest.warns(UserWarning):
        mod._dp_worker(torch.device('cpu'))          # built here: its random initialisation would shift the mask draws
    mod._dp_force_worker = False
    torch.manual_seed(3)
    out0 = reps[0](x[0:1])
    mod._dp_force_worker = True
    torch.manual_seed(4)
    out1 = reps[1](x[1:2])
    assert len(mod._dp_workers) == 1
    (crit(out0, labels[0:1]) + crit(out1, labels[1:2])).backward()
    got = torch.cat([p.grad.reshape(-1) for p in mod.parameters()])
    assert float((got - want).norm() / want.norm()) < 1e-5
    # eval-mode replicas give the master's logits
    mod.eval()
    reps = _fake_replicate(mod, 2)
    with torch.no_grad():
        ya, yb, ym = reps[0](x), reps[1](x), mod(x)
    assert torch.allclose(ya, ym, atol=1e-6) and torch.allclose(yb, ym, atol=1e-5)


def test_autograd_backward_keeps_arena_gradients(fp32_model):
    """loss.backward() through model(x) after a train_step must accumulate onto, not wipe, the gradients that the
    Parameters' .grad (views of the flat gradient arena) already hold."""
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    mod.train()
    x, labels = make_batch(1, 64, 64)
    masks = make_dropout_masks(ref, 1, torch.Generator().manual_seed(5))
    mod.train_step(x, labels, dropout_masks=masks)
    g1 = mod._gflat.clone()
    assert float(g1.abs().max()) > 0
    p0 = next(mod.parameters())
    assert p0.grad.data_ptr() == mod._gflat.data_ptr() + 4 * mod.arena.by_name[mod._param_names[0]].off
    mod.spec.dropout0 = mod.spec.dropout1 = 0.0      # the autograd path draws its own masks
    eng = mod.engine(1, 64, 64, True)
    eng.dropmask.fill_(1.0)
    torch.nn.functional.cross_entropy(mod(x), labels, ignore_index=255).backward()
    g_sum = mod._gflat.clone()
    mod._gflat_zero()
    torch.nn.functional.cross_entropy(mod(x), labels, ignore_index=255).backward()
    g2 = mod._gflat.clone()
    assert float((g_sum - (g1 + g2)).abs().max()) <= 1e-5 * float(g_sum.abs().max())


def test_input_gradient_fp32(fp32_model):
    """The gradient with respect to the image (the stem's data gradient): what autograd delivers to an image tensor that
    requires grad when the reference's loop runs loss.backward(), and what model.input_gradient returns after the fused
    train_step -- both against the oracle."""
    ref, mod = make_pair(TINY_CFG)
    mod.set_backend(kernel_model.ModelBackend())
    x, labels = make_batch(2, 32, 48)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    ref.train()
    xr = x.clone().requires_grad_(True)
    F.cross_entropy(ref(xr, masks), labels, ignore_index=255).backward()
    want = xr.grad
    assert float(want.abs().max()) > 0
    mod.train()
    mod.train_step(x, labels, dropout_masks=masks)
    got = mod.input_gradient(x)
    assert got.shape == x.shape
    assert float((got - want).norm() / want.norm()) < 1e-3
    # through autograd: the same engine state, dropout masks pinned
    mod.load_state_dict(ref.state_dict())
    eng = mod.engine(2, 32, 48, True)
    orig = eng.draw_dropout
    eng.draw_dropout = lambda generator=None, masks=None: orig(masks=_masks)
    _masks = masks
    xm = x.clone().requires_grad_(True)
    F.cross_entropy(mod(xm), labels, ignore_index=255).backward()
    assert xm.grad is not None and xm.grad.shape == x.shape
    # the oracle's running statistics moved in its forward pass; train-mode gradients do not depend on them
    assert float((xm.grad - want).norm() / want.norm()) < 1e-3
    # an image that does not require grad costs nothing and gets nothing
    x2 = x.clone()
    F.cross_entropy(mod(x2), labels, ignore_index=255).backward()
    assert x2.grad is None


def test_fused_stem_plan_matches_the_im2col_plan():
    """conv0 without a column matrix (stem7_fwd / stem7_wgrad) against im2col + GEMM + bn_bwd_apply + weight gradient:
    the same arithmetic with the same rounding points, so the bf16 kernel model must give the same loss and gradients."""
    cfg = dict(TINY_CFG)
    cfg.update(num_init_features=64, growth_rate=32, bn_size=1)      # 64 stem channels; block widths stay multiples of 8
    ref, mod = make_pair(cfg)
    x, labels = make_batch(2, 32, 48)
    masks = make_dropout_masks(ref, 2, torch.Generator().manual_seed(5))
    out = {}
    for fused in (True, False):
        m = copy.deepcopy(mod)
        m.fused_stem = fused
        m.set_backend(kernel_model.ModelBackend())
        m.train()
        loss = float(m.train_step(x, labels, dropout_masks=masks))
        eng = m.engine(2, 32, 48, True)
        assert eng.fused_stem == fused
        kinds = {k for k, _ in eng._fwd.ops} | {k for k, _ in eng._bwd.ops}
        assert ('stem7_fwd' in kinds and 'stem7_wgrad' in kinds) == fused and ('im2col' in kinds) == (not fused)
        out[fused] = (loss, {n: p.grad.clone() for n, p in m.named_parameters()}, m.input_gradient(x))
    assert abs(out[True][0] - out[False][0]) < 1e-6
    for n, g in out[False][1].items():
        assert torch.allclose(g, out[True][1][n], rtol=1e-4, atol=1e-7), n
    assert torch.allclose(out[True][2], out[False][2], rtol=1e-4, atol=1e-8)
