"""Generate the oracle's own golden vectors (tests/golden/oracle_tiny.pt).

The reference holds no fixtures for this path ("parity unpinned"), so these vectors pin the
oracle against accidental edits: seed-0 initialisation of a small DenseASPP, a seeded input,
and the eval-mode logits computed by this container's torch CPU build.
    python tests/golden/make_golden.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.dense_aspp_ref import DenseASPPRef  # noqa: E402

cfg = dict(bn_size=4, drop_rate=0, growth_rate=8, num_init_features=16, block_config=(2, 2, 2, 2),
           dropout0=0.1, dropout1=0.1, d_feature0=32, d_feature1=16)
seed = 0
torch.manual_seed(seed)
ref = DenseASPPRef(cfg).eval()
x = torch.randn(1, 3, 32, 48, generator=torch.Generator().manual_seed(7))
with torch.no_grad():
    y = ref(x)
out = dict(cfg=cfg, seed=seed, x=x, logits_eval=y, keys=list(ref.state_dict().keys()))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_tiny.pt')
torch.save(out, path)
print('wrote', path, os.path.getsize(path), 'bytes')
