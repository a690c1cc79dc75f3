"""Layer plan of DenseASPP derived from a Model_CFG dict (reference: cfgs/DenseASPP161.py:1-14).

The plan names every convolution / BatchNorm with the state_dict prefix it has in the
nn.Module tree and records the channel arithmetic of the feature slabs:
  dense block b:  slab of C_b channels, layer i reads channels [0, c0 + i*g) and appends g
  ASPP:           slab laid out [ASPP_24 | ASPP_18 | ASPP_12 | ASPP_6 | ASPP_3 | feature] so that
                  "concatenate in front" is a write just below the consumer's channel range
"""
ASPP_RATES = (3, 6, 12, 18, 24)
STEM_K = 147          # 3 * 7 * 7
STEM_KPAD = 160       # im2col row width (multiple of 32)
CLS_PAD = 32          # classifier rows after padding


class DenseLayerSpec:
    def __init__(self, prefix, cin, mid, growth, dil):
        self.prefix, self.cin, self.mid, self.growth, self.dil = prefix, cin, mid, growth, dil


class BlockSpec:
    def __init__(self, index, c0, layers, dil, pool_after, res=8):
        self.index, self.c0, self.layers, self.dil, self.pool_after = index, c0, layers, dil, pool_after
        self.res = res          # the block runs at 1/res of the input resolution
        self.C = c0 + sum(l.growth for l in layers)
        self.trans_prefix = 'features.transition%d' % index
        self.cout = self.C // 2


class AsppSpec:
    def __init__(self, prefix, rate, cin, d0, d1, bn_start, off_in, off_out):
        self.prefix, self.rate, self.cin, self.d0, self.d1 = prefix, rate, cin, d0, d1
        self.bn_start, self.off_in, self.off_out = bn_start, off_in, off_out


# MobileNetV2 trunk at output stride 8: (expansion t, channels c, repeats n, stride s, dilation)
MOBILENET_ROWS = (
    (1, 16, 1, 1, 1), (6, 24, 2, 2, 1), (6, 32, 3, 2, 1), (6, 64, 4, 1, 2),
    (6, 96, 3, 1, 4), (6, 160, 3, 1, 8), (6, 320, 1, 1, 16),
)
MOBILENET_STEM = 32
MOBILENET_STEM_K = 27     # 3 * 3 * 3
STEM3_KPAD = 32           # row width of the 3x3/2 im2col (stem weight gradient)


def is_mobilenet_cfg(cfg):
    """The MobileNet dict (reference: cfgs/MobileNetDenseASPP.py:1-8) has only the head keys."""
    return 'growth_rate' not in cfg and 'block_config' not in cfg


class IRBlockSpec:
    """One inverted-residual block: 1x1 expand -> depthwise 3x3 -> 1x1 linear, each followed by a BatchNorm."""

    def __init__(self, prefix, cin, cout, stride, dil, expand):
        self.prefix, self.cin, self.cout, self.stride, self.dil = prefix, cin, cout, stride, dil
        self.hid = cin * expand
        self.residual = stride == 1 and cin == cout


class NetSpec:
    kind = 'densenet'

    def __new__(cls, cfg=None, n_class=19, output_stride=8):
        if cls is NetSpec and cfg is not None and is_mobilenet_cfg(cfg):
            return object.__new__(MobileNetSpec)
        return object.__new__(cls)

    def _head(self, cfg, c):
        """ASPP stack + classifier plan on top of a c-channel trunk output."""
        self.c_feat = c
        d0, d1 = cfg['d_feature0'], cfg['d_feature1']
        self.d0, self.d1 = d0, d1
        self.C_aspp = c + 5 * d1
        self.aspp = []
        for i, r in enumerate(ASPP_RATES):
            off_in = (5 - i) * d1            # consumer range [off_in, C_aspp)
            self.aspp.append(AsppSpec('ASPP_%d' % r, r, c + i * d1, d0, d1, i > 0, off_in, off_in - d1))
        self.off_feat = 5 * d1
        self.dropout0, self.dropout1 = cfg['dropout0'], cfg['dropout1']

    def __init__(self, cfg, n_class=19, output_stride=8):
        if int(output_stride) not in (8, 16):
            raise ValueError('output_stride must be 8 or 16 (got %r)' % (output_stride,))
        self.cfg = dict(cfg)
        self.n_class = n_class
        self.output_stride = int(output_stride)
        fs = self.output_stride // 8          # 2: transition2 pools, blocks 3-4 run at 1/16 and norm5 is followed by a x2 upsample
        g, bn = cfg['growth_rate'], cfg['bn_size']
        if cfg.get('drop_rate', 0):
            raise NotImplementedError('dense-layer drop_rate > 0 is not implemented (all reference cfgs use 0)')
        c = cfg['num_init_features']
        self.c_stem = c
        dil = (1, 1, 2 // fs, 4 // fs)
        res = (4, 8, 8 * fs, 8 * fs)
        self.blocks = []
        for b, n in enumerate(cfg['block_config']):
            layers = [DenseLayerSpec('features.denseblock%d.denselayer%d' % (b + 1, i + 1), c + i * g, bn * g, g, dil[b])
                      for i in range(n)]
            blk = BlockSpec(b + 1, c, layers, dil[b], pool_after=(b == 0 or (b == 1 and fs == 2)), res=res[b])
            self.blocks.append(blk)
            c = blk.cout
        self._head(cfg, c)
        d0, d1 = self.d0, self.d1
        for v in [self.c_stem, g, bn * g, d0, d1, self.c_feat] + [b.cout for b in self.blocks]:
            if v % 8:
                raise ValueError('channel count %d is not a multiple of 8' % v)

    def conv_macs(self, height, width):
        macs = (height // 2) * (width // 2) * STEM_K * self.c_stem
        for blk in self.blocks:
            px = (height // blk.res) * (width // blk.res)
            for l in blk.layers:
                macs += px * l.cin * l.mid + px * l.mid * 9 * l.growth
            macs += px * blk.C * blk.cout
        px = (height // 8) * (width // 8)
        for a in self.aspp:
            macs += px * a.cin * a.d0 + px * a.d0 * 9 * a.d1
        macs += px * self.C_aspp * self.n_class
        return macs


class MobileNetSpec(NetSpec):
    """MobileNetV2 trunk (stride-2 3x3 stem + inverted residual blocks, output stride 8) + the DenseASPP head."""
    kind = 'mobilenet'

    def __init__(self, cfg, n_class=19, output_stride=8):
        if int(output_stride) != 8:
            raise NotImplementedError('the MI355X path implements output_stride=8 (got %r)' % (output_stride,))
        self.cfg = dict(cfg)
        self.n_class = n_class
        self.c_stem = MOBILENET_STEM
        self.stem_prefix = 'features.features.0'
        self.ir = []
        c = self.c_stem
        idx = 1
        for t, oup, n, s, d in MOBILENET_ROWS:
            for i in range(n):
                self.ir.append(IRBlockSpec('features.features.%d' % idx, c, oup, s if i == 0 else 1, d, t))
                c = oup
                idx += 1
        self.blocks = []
        self._head(cfg, c)
        for v in (self.d0, self.d1):
            if v % 8:
                raise ValueError('channel count %d is not a multiple of 8' % v)

    def conv_macs(self, height, width):
        h, w = height // 2, width // 2
        macs = h * w * MOBILENET_STEM_K * self.c_stem
        for b in self.ir:
            macs += h * w * b.cin * b.hid
            h, w = h // b.stride, w // b.stride
            macs += h * w * b.hid * 9 + h * w * b.hid * b.cout
        px = h * w
        for a in self.aspp:
            macs += px * a.cin * a.d0 + px * a.d0 * 9 * a.d1
        macs += px * self.C_aspp * self.n_class
        return macs
