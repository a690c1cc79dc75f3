from cfgs._schema import densenet_cfg

# DenseNet-169 trunk, ASPP 336/168 (reference: cfgs/DenseASPP169.py:1-14)
Model_CFG = densenet_cfg(32, 64, (6, 12, 32, 32), 336, 168, "./pretrained/densenet169.pth")
