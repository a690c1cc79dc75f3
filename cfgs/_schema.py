"""Builders for the DenseASPP hyper-parameter dictionaries.

Every ``cfgs/<Name>.py`` module exposes a module-level ``Model_CFG`` dict with the
key set the reference scripts pass to the model constructor
(reference: cfgs/DenseASPP161.py:1-14, cfgs/MobileNetDenseASPP.py:1-8; call site
graph_seg_test.py:129).  The values are pinned by tests/golden/cfg_snapshot.json.
"""


def densenet_cfg(growth_rate, num_init_features, block_config, d_feature0, d_feature1,
                 pretrained_path=None, dropout0=0.1, dropout1=0.1, bn_size=4, drop_rate=0):
    cfg = dict(bn_size=bn_size, drop_rate=drop_rate, growth_rate=growth_rate,
               num_init_features=num_init_features, block_config=tuple(block_config),
               dropout0=dropout0, dropout1=dropout1,
               d_feature0=d_feature0, d_feature1=d_feature1)
    if pretrained_path is not None:
        cfg['pretrained_path'] = pretrained_path
    return cfg


def mobilenet_cfg(d_feature0, d_feature1, pretrained_path, dropout0=0.1, dropout1=0.1):
    return dict(dropout0=dropout0, dropout1=dropout1, d_feature0=d_feature0,
                d_feature1=d_feature1, pretrained_path=pretrained_path)
