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
    def __init__(self, index, c0, layers, dil, pool_after):
        self.index, self.c0, self.layers, self.dil, self.pool_after = index, c0, layers, dil, pool_after
        self.C = c0 + sum(l.growth for l in layers)
        self.trans_prefix = 'features.transition%d' % index
        self.cout = self.C // 2


class AsppSpec:
    def __init__(self, prefix, rate, cin, d0, d1, bn_start, off_in, off_out):
        self.prefix, self.rate, self.cin, self.d0, self.d1 = prefix, rate, cin, d0, d1
        self.bn_start, self.off_in, self.off_out = bn_start, off_in, off_out


class NetSpec:
    def __init__(self, cfg, n_class=19, output_stride=8):
        if int(output_stride) != 8:
            raise NotImplementedError('the MI355X path implements output_stride=8 (got %r)' % (output_stride,))
        self.cfg = dict(cfg)
        self.n_class = n_class
        g, bn = cfg['growth_rate'], cfg['bn_size']
        if cfg.get('drop_rate', 0):
            raise NotImplementedError('dense-layer drop_rate > 0 is not implemented (all reference cfgs use 0)')
        c = cfg['num_init_features']
        self.c_stem = c
        dil = (1, 1, 2, 4)
        self.blocks = []
        for b, n in enumerate(cfg['block_config']):
            layers = [DenseLayerSpec('features.denseblock%d.denselayer%d' % (b + 1, i + 1), c + i * g, bn * g, g, dil[b])
                      for i in range(n)]
            blk = BlockSpec(b + 1, c, layers, dil[b], pool_after=(b == 0))
            self.blocks.append(blk)
            c = blk.cout
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
        for v in [self.c_stem, g, bn * g, d0, d1, self.c_feat] + [b.cout for b in self.blocks]:
            if v % 8:
                raise ValueError('channel count %d is not a multiple of 8' % v)

    def conv_macs(self, height, width):
        macs = (height // 2) * (width // 2) * STEM_K * self.c_stem
        px = (height // 4) * (width // 4)
        for blk in self.blocks:
            for l in blk.layers:
                macs += px * l.cin * l.mid + px * l.mid * 9 * l.growth
            macs += px * blk.C * blk.cout
            if blk.pool_after:
                px //= 4
        for a in self.aspp:
            macs += px * a.cin * a.d0 + px * a.d0 * 9 * a.d1
        macs += px * self.C_aspp * self.n_class
        return macs
