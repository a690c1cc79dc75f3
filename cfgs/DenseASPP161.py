from cfgs._schema import densenet_cfg

# DenseNet-161 trunk, ASPP 512/128 (reference: cfgs/DenseASPP161.py:1-14)
Model_CFG = densenet_cfg(48, 96, (6, 12, 36, 24), 512, 128, "./pretrained/densenet161.pth")
