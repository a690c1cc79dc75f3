"""CPU model of the libdsaspp kernel descriptors -- TEST INFRASTRUCTURE ONLY.

Interprets the descriptor lists that engine.Engine records (the same fields
include/dsaspp.h defines) with plain torch ops on CPU tensors.  Two uses:
  * `-m "not gpu"` tests run the whole forward/backward orchestration (slab layout,
    BatchNorm correction bookkeeping, gradient routing) against the oracle without a GPU;
  * `-m gpu` tests compare each HIP kernel with the same function on the same inputs.
Nothing in the product package imports this file.
"""
import torch
import torch.nn.functional as F

R = 16
bf16 = torch.bfloat16     # storage type of activations; tests may switch the model to fp32 (set_act_dtype)


def set_act_dtype(dt):
    global bf16
    bf16 = dt


def _view(ref, rows, cols, ld):
    """2-D strided view rows x cols (row stride ld) starting at a Ref."""
    if ref is None:
        return None
    return torch.as_strided(ref.t, (int(rows), int(cols)), (int(ld), 1), ref.off)


def _vec(ref, n):
    if ref is None:
        return None
    return torch.as_strided(ref.t, (int(n),), (1,), ref.off)


def _rbf(x):
    return x.to(bf16).to(torch.float32)


def _prologue(x, f, n_img, hw, cin):
    """x: [M, cin] fp32 -> normalised/activated/masked bf16-rounded fp32."""
    a = x
    if f['in_scale'] is not None:
        a = a * _vec(f['in_scale'], cin) + _vec(f['in_shift'], cin)
    if f['in_relu']:
        a = a.clamp_min(0)
    if f['in_relu'] == 2:
        a = a.clamp_max(6)
    if f['in_mask'] is not None:
        mk = _view(f['in_mask'], n_img, cin, f['in_mask_ld'])
        a = (a.view(n_img, hw, cin) * mk[:, None, :]).reshape(-1, cin)
    if f['in_scale'] is not None or f['in_relu'] or f['in_mask'] is not None:
        a = _rbf(a)
    return a


def op_conv(f):
    n, H, W, cin, cout, taps, dil = f['n_img'], f['H'], f['W'], f['cin'], f['cout'], f['taps'], f['dil']
    M, hw = n * H * W, H * W
    x = _view(f['x'], M, cin, f['ldx']).to(torch.float32)
    if f.get('gx_corr1') is not None:
        # gradient read-out prologue: what grad_extract computes from the slab slice, rounded to bf16; optional copy out
        xs = _view(f['gx_xs'], M, cin, f['gx_ldxs']).to(torch.float32)
        x = x - (_vec(f['gx_corr1'], cin) + ((xs - _vec(f['gx_mean'], cin)) * _vec(f['gx_rstd'], cin)) * _vec(f['gx_corr2'], cin))
        x = _rbf(x)
        if f.get('gx_dz') is not None:
            _view(f['gx_dz'], M, cin, f['gx_lddz']).copy_(x.to(bf16))
    a = _prologue(x, f, n, hw, cin)
    w = _view(f['w'], cout, taps * cin, taps * cin).to(torch.float32)
    if taps == 1:
        acc = a @ w.t()
    else:
        a4 = a.view(n, H, W, cin).permute(0, 3, 1, 2)
        w4 = w.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        acc = F.conv2d(a4, w4, padding=dil, dilation=dil).permute(0, 2, 3, 1).reshape(M, cout)
    v = acc
    if f['bias'] is not None:
        v = v + _vec(f['bias'], cout)
    if f['out_mask'] is not None:
        mk = _view(f['out_mask'], n, cout, f['out_mask_ld'])
        v = (v.view(n, hw, cout) * mk[:, None, :]).reshape(M, cout)
    ref = None
    if f['epi_relu'] or f['s_acc'] is not None:
        ref = _view(f['ref'], M, cout, f['ldref']).to(torch.float32)
    if f['epi_relu']:
        pre = ref * _vec(f['e_scale'], cout) + _vec(f['e_shift'], cout) if f['e_scale'] is not None else ref
        keep = (pre > 0) & (pre < 6) if f['epi_relu'] == 2 else pre > 0
        v = torch.where(keep, v, torch.zeros_like(v))
    if f['s_acc'] is not None:
        xhat = (ref - _vec(f['e_mean'], cout)) * _vec(f['e_rstd'], cout)
        sld = f.get('s_ld', 0) or cout
        acc2 = _view(f['s_acc'], R, sld + cout, 2 * sld)
        acc2[0, :cout] += v.sum(0)
        acc2[0, sld:sld + cout] += (v * xhat).sum(0)
    if f['g'] is not None:
        g = _view(f['g'], M, cout, f['ldg'])
        t = v * _vec(f['e_scale'], cout) if f['e_scale'] is not None else v
        for k in range(1, f.get('n_src', 0) or 1):
            xs = _view(f['x%d' % k], M, cin, f['ldx']).to(torch.float32)
            ws = _view(f['w%d' % k], cout, cin, cin).to(torch.float32)
            vs = xs @ ws.t()
            pre = ref * _vec(f['es%d' % k], cout) + _vec(f['eh%d' % k], cout)
            vs = torch.where(pre > 0, vs, torch.zeros_like(vs))
            if f['sa%d' % k] is not None:
                sld = f['sl%d' % k] or cout
                acck = _view(f['sa%d' % k], R, sld + cout, 2 * sld)
                acck[0, :cout] += vs.sum(0)
                acck[0, sld:sld + cout] += (vs * xhat).sum(0)
            t = t + vs * _vec(f['es%d' % k], cout)
        if f['g_accumulate']:
            t = g.to(torch.float32) + t
        g.copy_(t.to(g.dtype))
    else:
        if f['y_f32']:
            y = _view(f['y'], M, cout, f['ldy'])
            y.copy_(v)
            stored = v
        else:
            y = _view(f['y'], M, cout, f['ldy'])
            y.copy_(v.to(bf16))
            stored = _rbf(v)
        if f['stat_acc'] is not None:
            st = _view(f['stat_acc'], R, 2 * cout, 2 * cout)
            st[0, :cout] += stored.sum(0)
            st[0, cout:] += (stored * stored).sum(0)


def op_wgrad(f):
    n, H, W, cin, cout, taps, dil = f['n_img'], f['H'], f['W'], f['cin'], f['cout'], f['taps'], f['dil']
    M, hw = n * H * W, H * W
    if f.get('n_src', 0) >= 2:
        # several 1x1 layers reading the same input through their own scale/shift tables: one source at a time
        assert taps == 1 and f['in_mask'] is None
        srcs = [(f['dz'], f['dw'], f['in_scale'], f['in_shift'], cin)]
        for k in range(1, f['n_src']):
            srcs.append(tuple(f['%s%d' % (nm, k)] for nm in ('dz', 'dw', 'sc', 'sh', 'cin')))
        for dz_r, dw_r, sc_r, sh_r, ci in srcs:
            g = dict(f, n_src=0, dz=dz_r, dw=dw_r, in_scale=sc_r, in_shift=sh_r, cin=ci)
            op_wgrad(g)
        return
    x = _view(f['x'], M, cin, f['ldx']).to(torch.float32)
    a = _prologue(x, f, n, hw, cin)
    dz = _view(f['dz'], M, cout, f['lddz']).to(torch.float32)
    dw = _view(f['dw'], cout, taps * cin, taps * cin)
    if taps == 1:
        dw += dz.t() @ a
    else:
        a4 = a.view(n, H, W, cin).permute(0, 3, 1, 2)
        dz4 = dz.view(n, H, W, cout).permute(0, 3, 1, 2)
        gw = torch.nn.grad.conv2d_weight(a4, (cout, cin, 3, 3), dz4, padding=dil, dilation=dil)
        dw += gw.permute(0, 2, 3, 1).reshape(cout, 9 * cin)


def op_bn_finalize(f):
    C = f['C']
    if f['use_running']:
        mean = _vec(f['running_mean'], C).clone()
        var = _vec(f['running_var'], C).clone()
    else:
        mean_t, var_t, rstd_t = _vec(f['mean'], C), _vec(f['var'], C), _vec(f['rstd'], C)
        if f['acc'] is not None and f['acc_cn'] > 0:
            cn, c0 = f['acc_cn'], f['acc_c0']
            acc = _view(f['acc'], R, 2 * cn, 2 * cn)
            s, q = acc[:, :cn].sum(0), acc[:, cn:].sum(0)
            acc.zero_()
            m = s / f['count']
            v = (q / f['count'] - m * m).clamp_min(0)
            mean_t[c0:c0 + cn] = m
            var_t[c0:c0 + cn] = v
            rstd_t[c0:c0 + cn] = torch.rsqrt(v + f['eps'])
        mean, var = mean_t.clone(), var_t.clone()
        if f['running_mean'] is not None:
            cnt = f['count']
            unb = var * cnt / (cnt - 1) if cnt > 1 else var
            rm, rv = _vec(f['running_mean'], C), _vec(f['running_var'], C)
            rm.mul_(1 - f['momentum']).add_(f['momentum'] * mean)
            rv.mul_(1 - f['momentum']).add_(f['momentum'] * unb)
            if f['num_batches_tracked'] is not None:
                f['num_batches_tracked'].t.add_(1)
    if f['scale'] is not None:
        rstd = torch.rsqrt(var + f['eps'])
        g = _vec(f['gamma'], C) if f['gamma'] is not None else torch.ones(C)
        b = _vec(f['beta'], C) if f['beta'] is not None else torch.zeros(C)
        sc = g * rstd
        _vec(f['scale'], C).copy_(sc)
        _vec(f['shift'], C).copy_(b - mean * sc)


def op_bn_bwd_finalize(f):
    C = f['C']
    ald = f.get('acc_ld', 0) or C
    acc = _view(f['acc'], R, ald + C, 2 * ald)
    s1, s2 = acc[:, :C].sum(0), acc[:, ald:ald + C].sum(0)
    acc[:, :C] = 0
    acc[:, ald:ald + C] = 0
    if f['dbeta'] is not None:
        _vec(f['dbeta'], C).add_(s1)
    if f['dgamma'] is not None:
        _vec(f['dgamma'], C).add_(s2)
    sc = _vec(f['scale'], C)
    inv = 1.0 / f['count']
    if f['mode'] == 0:
        Bc = -sc * _vec(f['rstd'], C) * s2 * inv
        _vec(f['coefA'], C).copy_(sc)
        _vec(f['coefB'], C).copy_(Bc)
        _vec(f['coefC'], C).copy_(-sc * s1 * inv - Bc * _vec(f['mean'], C))
    else:
        _vec(f['corr1'], C).add_(sc * s1 * inv)
        _vec(f['corr2'], C).add_(sc * s2 * inv)


def op_bn_act(f):
    M, C = f['M'], f['C']
    x = _view(f['x'], M, C, f['ldx']).to(torch.float32)
    v = x
    if f['scale'] is not None:
        v = v * _vec(f['scale'], C) + _vec(f['shift'], C)
    if f.get('res') is not None:
        v = v + _view(f['res'], M, C, f['ldres']).to(torch.float32)
    if f['relu']:
        v = v.clamp_min(0)
    if f['relu'] == 2:
        v = v.clamp_max(6)
    _view(f['y'], M, C, f['ldy']).copy_(v.to(bf16))
    if f['stat_acc'] is not None:
        s = _rbf(v)
        st = _view(f['stat_acc'], R, 2 * C, 2 * C)
        st[0, :C] += s.sum(0)
        st[0, C:] += (s * s).sum(0)


def op_grad_extract(f):
    M, C = f['M'], f['C']
    v = _view(f['g'], M, C, f['ldg']).to(torch.float32).clone()
    if f['corr1'] is not None:
        xs = _view(f['xs'], M, C, f['ldxs']).to(torch.float32)
        xhat = (xs - _vec(f['s_mean'], C)) * _vec(f['s_rstd'], C)
        v = v - _vec(f['corr1'], C) - xhat * _vec(f['corr2'], C)
    if f['mask'] is not None:
        n = M // f['hw']
        mk = _view(f['mask'], n, C, f['mask_ld'])
        v = (v.view(n, f['hw'], C) * mk[:, None, :]).reshape(M, C)
    y = None
    if f['relu'] or f['s_acc'] is not None:
        y = _view(f['yref'], M, C, f['ldyref']).to(torch.float32)
    if f['relu']:
        pre = y * _vec(f['r_scale'], C) + _vec(f['r_shift'], C) if f['r_scale'] is not None else y
        keep = (pre > 0) & (pre < 6) if f['relu'] == 2 else pre > 0
        v = torch.where(keep, v, torch.zeros_like(v))
    _view(f['dz'], M, C, f['lddz']).copy_(v if f['dz_f32'] else v.to(bf16))
    if f['s_acc'] is not None:
        xh = (y - _vec(f['r_mean'], C)) * _vec(f['r_rstd'], C)
        acc = _view(f['s_acc'], R, 2 * C, 2 * C)
        acc[0, :C] += v.sum(0)
        acc[0, C:] += (v * xh).sum(0)
    elif f['colsum'] is not None:
        _vec(f['colsum'], C).add_(v.sum(0))


def op_bn_bwd_apply(f):
    M, C = f['M'], f['C']
    dr = _view(f['dr'], M, C, f['lddr']).to(torch.float32)
    y = _view(f['y'], M, C, f['ldy']).to(torch.float32)
    o = _vec(f['coefA'], C) * dr + (_vec(f['coefB'], C) * y + _vec(f['coefC'], C))
    _view(f['dy'], M, C, f['lddy']).copy_(o.to(bf16))
    if f['colsum'] is not None:
        _vec(f['colsum'], C).add_(o.sum(0))


def op_avgpool_fwd(f):
    n, H, W, C = f['n_img'], f['H'], f['W'], f['C']
    x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32).view(n, H, W, C)
    o = 0.25 * ((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + (x[:, 1::2, 0::2] + x[:, 1::2, 1::2]))
    o = o.reshape(-1, C)
    _view(f['y'], o.shape[0], C, f['ldy']).copy_(o.to(bf16))
    if f['stat_acc'] is not None:
        s = _rbf(o)
        st = _view(f['stat_acc'], R, 2 * C, 2 * C)
        st[0, :C] += s.sum(0)
        st[0, C:] += (s * s).sum(0)


def op_avgpool_bwd(f):
    n, H, W, C = f['n_img'], f['H'], f['W'], f['C']
    dy = _view(f['y'], n * (H // 2) * (W // 2), C, f['ldy']).to(torch.float32).view(n, H // 2, W // 2, C)
    dx = (0.25 * dy).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1, C)
    _view(f['x'], n * H * W, C, f['ldx']).copy_(dx.to(bf16))


def op_im2col(f):
    n, H, W = f['n_img'], f['H'], f['W']
    x = torch.as_strided(f['x'].t, (n, 3, H, W), (3 * H * W, H * W, W, 1), f['x'].off)
    cols = F.unfold(x, 7, padding=3, stride=2)              # n, 3*49 (ci, ky, kx), L
    L = cols.shape[-1]
    cols = cols.view(n, 3, 49, L).permute(0, 3, 2, 1).reshape(n * L, 147)   # (ky,kx) major, ci minor
    out = _view(f['col'], n * L, f['ldcol'], f['ldcol'])
    out.zero_()
    out[:, :147] = cols.to(bf16)


def _stem7_cols(f):
    n, H, W = f['n_img'], f['H'], f['W']
    x = torch.as_strided(f['x'].t, (n, 3, H, W), (3 * H * W, H * W, W, 1), f['x'].off)
    cols = F.unfold(x, 7, padding=3, stride=2)
    L = cols.shape[-1]
    return _rbf(cols.view(n, 3, 49, L).permute(0, 3, 2, 1).reshape(n * L, 147))     # (ky, kx) major, ci minor; bf16 values


def op_stem7_fwd(f):
    cout = f['cout']
    cols = _stem7_cols(f)
    w = _view(f['w'], cout, 147, f['ldw']).to(torch.float32)
    y = (cols @ w.t()).to(bf16)
    _view(f['y'], cols.shape[0], cout, f['ldy']).copy_(y)
    if f['stat_acc'] is not None:
        acc = _view(f['stat_acc'], R, 2 * cout, 2 * cout)
        yf = y.to(torch.float32)
        acc[0, :cout] += yf.sum(0)
        acc[0, cout:] += (yf * yf).sum(0)


def op_stem7_wgrad(f):
    cout = f['cout']
    cols = _stem7_cols(f)
    M = cols.shape[0]
    dr = _view(f['dr'], M, cout, f['lddr']).to(torch.float32)
    y = _view(f['yref'], M, cout, f['ldyref']).to(torch.float32)
    dy = _rbf(_vec(f['coefA'], cout) * dr + (_vec(f['coefB'], cout) * y + _vec(f['coefC'], cout)))
    _view(f['dw'], cout, 147, f['lddw']).add_(dy.t() @ cols)


def op_col2im(f):
    n, H, W, ks = f['n_img'], f['H'], f['W'], f['ks']
    pad = 3 if ks == 7 else 1
    L = (H // 2) * (W // 2)
    k = ks * ks * 3
    dcol = _view(f['dcol'], n * L, k, f['ldcol']).to(torch.float32)          # (ky, kx) major, ci minor
    cols = dcol.view(n, L, ks * ks, 3).permute(0, 3, 2, 1).reshape(n, k, L)   # unfold's (ci, ky, kx) rows
    dx = F.fold(cols, (H, W), ks, padding=pad, stride=2)
    out = torch.as_strided(f['dx'].t, (n, 3, H, W), (3 * H * W, H * W, W, 1), f['dx'].off)
    if f['accumulate']:
        out.add_(dx)
    else:
        out.copy_(dx)


def op_stem3(f):
    n, H, W, cout = f['n_img'], f['H'], f['W'], f['cout']
    x = torch.as_strided(f['x'].t, (n, 3, H, W), (3 * H * W, H * W, W, 1), f['x'].off)
    w = _view(f['w'], cout, 27, 27).to(torch.float32).view(cout, 3, 3, 3).permute(0, 3, 1, 2)    # stored (ky, kx, ci)
    y = F.conv2d(_rbf(x), w, stride=2, padding=1)
    y = y.permute(0, 2, 3, 1).reshape(-1, cout)
    _view(f['y'], y.shape[0], cout, f['ldy']).copy_(y.to(bf16))


def op_dwconv(f):
    n, H, W, C, st, dil = f['n_img'], f['H'], f['W'], f['C'], f['stride'], f['dil']
    x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32)
    a = x
    if f['in_scale'] is not None:
        a = a * _vec(f['in_scale'], C) + _vec(f['in_shift'], C)
    if f['in_relu']:
        a = a.clamp_min(0)
    if f['in_relu'] == 2:
        a = a.clamp_max(6)
    if f['in_scale'] is not None or f['in_relu']:
        a = _rbf(a)
    a4 = a.view(n, H, W, C).permute(0, 3, 1, 2)
    w = _view(f['w'], C, 9, 9).to(torch.float32).view(C, 1, 3, 3)
    y = F.conv2d(a4, w, stride=st, padding=dil, dilation=dil, groups=C)
    y = y.permute(0, 2, 3, 1).reshape(-1, C)
    _view(f['y'], y.shape[0], C, f['ldy']).copy_(y.to(bf16))


def _dw_act(f, x, C):
    a = x
    if f['in_scale'] is not None:
        a = a * _vec(f['in_scale'], C) + _vec(f['in_shift'], C)
    pre = a
    if f['in_relu']:
        a = a.clamp_min(0)
    if f['in_relu'] == 2:
        a = a.clamp_max(6)
    if f['in_scale'] is not None or f['in_relu']:
        a = _rbf(a)
    return a, pre


def op_dw_bwd_data(f):
    n, H, W, C, st, dil = f['n_img'], f['H'], f['W'], f['C'], f['stride'], f['dil']
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    dy = _view(f['dy'], n * Ho * Wo, C, f['lddy']).to(torch.float32).view(n, Ho, Wo, C).permute(0, 3, 1, 2)
    w = _view(f['w'], C, 9, 9).to(torch.float32).view(C, 1, 3, 3)
    da = torch.nn.grad.conv2d_input((n, C, H, W), w, dy, stride=st, padding=dil, dilation=dil, groups=C)
    v = da.permute(0, 2, 3, 1).reshape(n * H * W, C)
    x = None
    if f['in_relu'] or f['s_acc'] is not None:
        x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32)
    if f['in_relu']:
        _, pre = _dw_act(f, x, C)
        keep = (pre > 0) & (pre < 6) if f['in_relu'] == 2 else pre > 0
        v = torch.where(keep, v, torch.zeros_like(v))
    _view(f['dx'], n * H * W, C, f['lddx']).copy_(v if f['dx_f32'] else v.to(bf16))
    if f['s_acc'] is not None:
        xh = (x - _vec(f['e_mean'], C)) * _vec(f['e_rstd'], C)
        acc = _view(f['s_acc'], R, 2 * C, 2 * C)
        acc[0, :C] += v.sum(0)
        acc[0, C:] += (v * xh).sum(0)


def op_dw_wgrad(f):
    n, H, W, C, st, dil = f['n_img'], f['H'], f['W'], f['C'], f['stride'], f['dil']
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32)
    a, _ = _dw_act(f, x, C)
    a4 = a.view(n, H, W, C).permute(0, 3, 1, 2)
    dy = _view(f['dy'], n * Ho * Wo, C, f['lddy']).to(torch.float32).view(n, Ho, Wo, C).permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(a4, (C, 1, 3, 3), dy, stride=st, padding=dil, dilation=dil, groups=C)
    _view(f['dw'], C, 9, 9).add_(gw.reshape(C, 9))


def op_stem3_col(f):
    n, H, W = f['n_img'], f['H'], f['W']
    x = torch.as_strided(f['x'].t, (n, 3, H, W), (3 * H * W, H * W, W, 1), f['x'].off)
    cols = F.unfold(x, 3, padding=1, stride=2)               # n, 3*9 (ci, ky, kx), L
    L = cols.shape[-1]
    cols = cols.view(n, 3, 9, L).permute(0, 3, 2, 1).reshape(n * L, 27)       # (ky,kx) major, ci minor
    out = _view(f['col'], n * L, f['ldcol'], f['ldcol'])
    out.zero_()
    out[:, :27] = cols.to(bf16)


def op_up2_fwd(f):
    n, h, w, C = f['n_img'], f['h'], f['w'], f['C']
    x = _view(f['x'], n * h * w, C, f['ldx']).to(torch.float32).view(n, h, w, C).permute(0, 3, 1, 2)
    v = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False).permute(0, 2, 3, 1).reshape(-1, C)
    if f['scale'] is not None:
        v = v * _vec(f['scale'], C) + _vec(f['shift'], C)
    if f['relu']:
        v = v.clamp_min(0)
    _view(f['y'], v.shape[0], C, f['ldy']).copy_(v.to(bf16))
    if f['stat_acc'] is not None:
        s = _rbf(v)
        st = _view(f['stat_acc'], R, 2 * C, 2 * C)
        st[0, :C] += s.sum(0)
        st[0, C:] += (s * s).sum(0)


def op_up2_bwd(f):
    n, h, w, C = f['n_img'], f['h'], f['w'], f['C']
    dy = _view(f['y'], n * 4 * h * w, C, f['ldy']).to(torch.float32).view(n, 2 * h, 2 * w, C).permute(0, 3, 1, 2)
    with torch.enable_grad():
        x0 = torch.zeros(n, C, h, w, requires_grad=True)
        F.interpolate(x0, scale_factor=2, mode='bilinear', align_corners=False).backward(dy)
    dx = x0.grad.permute(0, 2, 3, 1).reshape(-1, C)
    _view(f['dx'], n * h * w, C, f['lddx']).copy_(dx if f['dx_f32'] else dx.to(bf16))
    if f['s_acc'] is not None:
        yv = _view(f['x'], n * h * w, C, f['ldx']).to(torch.float32)
        xh = (yv - _vec(f['mean'], C)) * _vec(f['rstd'], C)
        acc = _view(f['s_acc'], R, 2 * C, 2 * C)
        acc[0, :C] += dx.sum(0)
        acc[0, C:] += (dx * xh).sum(0)


def op_maxpool_fwd(f):
    n, H, W, C = f['n_img'], f['H'], f['W'], f['C']
    x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32)
    v = (x * _vec(f['scale'], C) + _vec(f['shift'], C)).clamp_min(0)
    v4 = v.view(n, H, W, C).permute(0, 3, 1, 2)
    o, idx = F.max_pool2d(v4, 3, stride=2, padding=1, return_indices=True)
    Ho, Wo = o.shape[2], o.shape[3]
    o2 = o.permute(0, 2, 3, 1).reshape(-1, C)
    _view(f['y'], o2.shape[0], C, f['ldy']).copy_(o2.to(bf16))
    if f['argmax'] is not None:
        ih, iw = idx // W, idx % W
        ho = torch.arange(Ho).view(1, 1, Ho, 1)
        wo = torch.arange(Wo).view(1, 1, 1, Wo)
        t = (ih - (2 * ho - 1)) * 3 + (iw - (2 * wo - 1))
        _view(f['argmax'], o2.shape[0], C, C).copy_(t.permute(0, 2, 3, 1).reshape(-1, C).to(torch.uint8))
    if f['stat_acc'] is not None:
        s = _rbf(o2)
        st = _view(f['stat_acc'], R, 2 * C, 2 * C)
        st[0, :C] += s.sum(0)
        st[0, C:] += (s * s).sum(0)


def op_maxpool_bwd(f):
    n, H, W, C = f['n_img'], f['H'], f['W'], f['C']
    Ho, Wo = H // 2, W // 2
    x = _view(f['x'], n * H * W, C, f['ldx']).to(torch.float32)
    dy = _view(f['dy'], n * Ho * Wo, C, f['lddy']).to(torch.float32).view(n, Ho, Wo, C)
    am = _view(f['argmax'], n * Ho * Wo, C, C).to(torch.int64).view(n, Ho, Wo, C)
    ho = torch.arange(Ho).view(1, Ho, 1, 1)
    wo = torch.arange(Wo).view(1, 1, Wo, 1)
    ih = 2 * ho - 1 + am // 3
    iw = 2 * wo - 1 + am % 3
    g = torch.zeros(n, H * W, C)
    g.scatter_add_(1, (ih * W + iw).view(n, Ho * Wo, C), dy.reshape(n, Ho * Wo, C))
    g = g.view(n * H * W, C)
    pre = x * _vec(f['scale'], C) + _vec(f['shift'], C)
    g = torch.where(pre > 0, g, torch.zeros_like(g))
    _view(f['dr'], n * H * W, C, f['lddr']).copy_(g if f['dr_f32'] else g.to(bf16))
    if f['s_acc'] is not None:
        xh = (x - _vec(f['mean'], C)) * _vec(f['rstd'], C)
        acc = _view(f['s_acc'], R, 2 * C, 2 * C)
        acc[0, :C] += g.sum(0)
        acc[0, C:] += (g * xh).sum(0)


def op_upce(f):
    n, h, w, nc, ldl = f['n_img'], f['h'], f['w'], f['n_class'], f['ldl']
    lg = _view(f['logits'], n * h * w, ldl, ldl)[:, :nc].view(n, h, w, nc).permute(0, 3, 1, 2).clone().requires_grad_(True)
    lab = torch.as_strided(f['labels'].t, (n, 8 * h, 8 * w), (64 * h * w, 8 * w, 1), f['labels'].off)
    full = F.interpolate(lg, scale_factor=8, mode='bilinear', align_corners=False)
    loss_sum = F.cross_entropy(full, lab, ignore_index=f['ignore_index'], reduction='sum')
    loss_sum.backward()
    cnt = (lab != f['ignore_index']).sum()
    acc = _vec(f['accum'], 1024)               # 32 partial (loss, count) pairs, stride 32 floats; the model uses the first
    acc[0] += loss_sum.detach()
    acc[1] += cnt
    dl = _view(f['dlogits'], n * h * w, ldl, ldl)
    dl[:, :nc] += lg.grad.permute(0, 2, 3, 1).reshape(-1, nc)


def op_ce_finalize(f):
    n, h, w, nc, ldl = f['n_img'], f['h'], f['w'], f['n_class'], f['ldl']
    acc = _vec(f['accum'], 1024).view(32, 32).sum(0)
    cnt = float(acc[1])
    inv = f['grad_scale'] / cnt if cnt > 0 else 0.0
    if f['loss'] is not None:
        _vec(f['loss'], 1)[0] = float(acc[0]) / cnt if cnt > 0 else 0.0
    dl = _view(f['dlogits'], n * h * w, ldl, ldl)
    dl[:, :nc] *= inv
    dl[:, nc:] = 0
    if f['dlogits_bf16'] is not None:
        ob = _view(f['dlogits_bf16'], n * h * w, f['lddb'], f['lddb'])
        ob.zero_()
        ob[:, :ldl] = dl.to(bf16)
    if f['dbias'] is not None:
        _vec(f['dbias'], nc).add_(dl[:, :nc].sum(0))


def op_up_fwd(f):
    n, h, w, nc, ldl = f['n_img'], f['h'], f['w'], f['n_class'], f['ldl']
    lg = _view(f['logits'], n * h * w, ldl, ldl)[:, :nc].view(n, h, w, nc).permute(0, 3, 1, 2)
    full = F.interpolate(lg, scale_factor=8, mode='bilinear', align_corners=False)
    torch.as_strided(f['out'].t, full.shape, full.contiguous().stride(), f['out'].off).copy_(full)


def op_up_argmax(f):
    n, h, w, nc, ldl = f['n_img'], f['h'], f['w'], f['n_class'], f['ldl']
    lg = _view(f['logits'], n * h * w, ldl, ldl)[:, :nc].view(n, h, w, nc).permute(0, 3, 1, 2)
    full = F.interpolate(lg, scale_factor=8, mode='bilinear', align_corners=False)
    am = full.argmax(1)
    torch.as_strided(f['argmax'].t, am.shape, am.contiguous().stride(), f['argmax'].off).copy_(am)


def op_up_bwd(f):
    n, h, w, nc, ldl = f['n_img'], f['h'], f['w'], f['n_class'], f['ldl']
    g = torch.as_strided(f['out'].t, (n, nc, 8 * h, 8 * w), (nc * 64 * h * w, 64 * h * w, 8 * w, 1), f['out'].off)
    lg = torch.zeros(n, nc, h, w, requires_grad=True)
    full = F.interpolate(lg, scale_factor=8, mode='bilinear', align_corners=False)
    full.backward(g)
    dl = _view(f['logits'], n * h * w, ldl, ldl)
    dl[:, :nc] = lg.grad.permute(0, 2, 3, 1).reshape(-1, nc)


def op_memset(f):
    t = f['dst'].t
    nel = f['n'] // t.element_size()
    assert f['value'] == 0
    _vec(f['dst'], nel).zero_()


def op_copy2d(f):
    src = _view(f['src'], f['rows'], f['cols'], f['lds'])
    dst = _view(f['dst'], f['rows'], f['dst_cols'], f['ldd'])
    if f['accumulate']:
        dst[:, :f['cols']] += src
    else:
        dst.zero_()
        dst[:, :f['cols']] = src


OPS = {
    'conv': op_conv, 'wgrad': op_wgrad, 'bn_finalize': op_bn_finalize, 'bn_bwd_finalize': op_bn_bwd_finalize,
    'bn_act': op_bn_act, 'grad_extract': op_grad_extract, 'bn_bwd_apply': op_bn_bwd_apply,
    'avgpool_fwd': op_avgpool_fwd, 'avgpool_bwd': op_avgpool_bwd, 'im2col': op_im2col,
    'maxpool_fwd': op_maxpool_fwd, 'maxpool_bwd': op_maxpool_bwd, 'upce': op_upce, 'ce_finalize': op_ce_finalize,
    'up_fwd': op_up_fwd, 'up_bwd': op_up_bwd, 'up_argmax': op_up_argmax, 'memset': op_memset, 'copy2d': op_copy2d,
    'stem3': op_stem3, 'dwconv': op_dwconv, 'up2_fwd': op_up2_fwd, 'up2_bwd': op_up2_bwd,
    'dw_bwd_data': op_dw_bwd_data, 'dw_wgrad': op_dw_wgrad, 'stem3_col': op_stem3_col, 'col2im': op_col2im, 'stem7_fwd': op_stem7_fwd, 'stem7_wgrad': op_stem7_wgrad,
}


class ModelProgram:
    def __init__(self, oplist):
        self.ops = list(oplist)

    def run(self, stream=None, first=0, last=-1):
        ops = self.ops[first:(None if last < 0 else last)]
        with torch.no_grad():
            for kind, f in ops:
                if kind in ('upce', 'up_bwd'):
                    with torch.enable_grad():
                        OPS[kind](f)
                else:
                    OPS[kind](f)

    def __len__(self):
        return len(self.ops)


class ModelBackend:
    """Drop-in for backend.HipBackend that interprets descriptors on the CPU (tests only)."""
    stat_replicas = R

    @property
    def act_dtype(self):
        return bf16

    @staticmethod
    def has_op(kind):
        return kind in OPS

    def compile(self, oplist):
        return ModelProgram(oplist)

    def call(self, kind, stream=None, **fields):
        ModelProgram([(kind, fields)]).run()

    def refresh_weights(self, model, cast=True):
        ar = model.arena
        model._w16.copy_(model._flat.to(bf16))
        if 'features.conv0.weight' in ar.by_name:
            e0 = ar.by_name['features.conv0.weight']
            w0 = model._flat[e0.off:e0.off + e0.rows * 147].view(e0.rows, 147)
            p = model._w0_16.view(e0.rows, -1)
            p.zero_()
            p[:, :147] = w0.to(bf16)
        for e in ar.entries:
            if e.kind != 'conv':
                continue
            src = model._w16[e.off:e.off + e.numel].view(e.rows, e.taps, e.cin)
            dst = src.flip(1).permute(2, 1, 0).contiguous()       # [cin][taps-1-t][cout]
            model._wt16[e.toff:e.toff + e.numel] = dst.reshape(-1)
