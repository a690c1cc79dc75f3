from cfgs._schema import densenet_cfg

# DenseNet-121 trunk, ASPP 128/64 (reference: cfgs/DenseASPP121.py:1-14)
Model_CFG = densenet_cfg(32, 64, (6, 12, 24, 16), 128, 64, "./pretrained/densenet121.pth")
