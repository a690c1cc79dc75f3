"""Generate tests/golden/cfg_snapshot.json by importing the reference's cfg modules.

Run in the build container only (needs /root/reference):
    python tests/golden/make_cfg_snapshot.py
The JSON holds data (the five Model_CFG dicts), not source text.
"""
import importlib.util
import json
import os

REF = '/root/reference/cfgs'
out = {}
for name in ['DenseASPP121', 'DenseASPP161', 'DenseASPP169', 'DenseASPP201', 'MobileNetDenseASPP']:
    spec = importlib.util.spec_from_file_location('refcfg_' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = dict(mod.Model_CFG)
    for k, v in cfg.items():
        if isinstance(v, tuple):
            cfg[k] = list(v)
    out[name] = cfg
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cfg_snapshot.json')
with open(path, 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
print('wrote', path)
