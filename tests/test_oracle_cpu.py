"""CPU tests of the oracle itself and of the pinned configuration values."""
import json
import os

import torch

from oracle.dense_aspp_ref import DenseASPPRef, conv_macs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_cfg_values_match_reference_snapshot():
    snap = json.load(open(os.path.join(GOLD, 'cfg_snapshot.json')))
    import importlib
    for name, want in snap.items():
        cfg = importlib.import_module('cfgs.' + name).Model_CFG
        got = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
        assert got == want, name


def test_macs_and_params_match_hand_computed_figures():
    from cfgs import DenseASPP121, DenseASPP161, DenseASPP169, DenseASPP201
    assert abs(conv_macs(DenseASPP161.Model_CFG, 1024, 2048) / 1e9 - 1239.5) < 0.1
    assert abs(conv_macs(DenseASPP121.Model_CFG, 512, 1024) / 1e9 - 77.2) < 0.1
    assert abs(conv_macs(DenseASPP169.Model_CFG, 1024, 2048) / 1e9 - 638.3) < 0.1
    assert abs(conv_macs(DenseASPP201.Model_CFG, 1024, 2048) / 1e9 - 970.7) < 0.1
    m = DenseASPPRef(DenseASPP121.Model_CFG)
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 8.28) < 0.01


def test_oracle_golden_vectors():
    g = torch.load(os.path.join(GOLD, 'oracle_tiny.pt'), weights_only=True)
    torch.manual_seed(g['seed'])
    ref = DenseASPPRef(g['cfg'])
    assert list(ref.state_dict().keys()) == g['keys']
    ref.eval()
    with torch.no_grad():
        y = ref(g['x'])
    assert torch.allclose(y, g['logits_eval'], atol=1e-5, rtol=1e-4)
