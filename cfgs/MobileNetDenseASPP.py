from cfgs._schema import mobilenet_cfg

# MobileNetV2 trunk, ASPP 128/64 (reference: cfgs/MobileNetDenseASPP.py:1-8)
Model_CFG = mobilenet_cfg(128, 64, "./pretrained/mobilenetv2.pth.tar")
