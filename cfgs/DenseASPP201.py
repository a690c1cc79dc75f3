from cfgs._schema import densenet_cfg

# DenseNet-201 trunk, ASPP 480/240; the reference dict carries no pretrained_path
# (reference: cfgs/DenseASPP201.py:1-12)
Model_CFG = densenet_cfg(32, 64, (6, 12, 48, 32), 480, 240)
