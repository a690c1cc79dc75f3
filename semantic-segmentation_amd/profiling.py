"""Kernel classes, their algorithmic work, and live HIP-event timing of a recorded program.

The convolution entry points dispatch on the output-channel count to a few tile instantiations;
`kernel_class` mirrors that dispatch so that timing groups coincide with the kernel symbols a
profiler shows.  Algorithmic work: 2*M*cout*taps*cin flops for a convolution or weight-gradient
launch; bytes for the fused upsample + cross-entropy kernel (low-res logits read + labels read +
low-res gradient written).
"""
import os

CLASSES = ['untimed',
           'conv_kernel<128,192,2,4>', 'conv_kernel<128,128,2,4>', 'conv_kernel<256,48,4,1>',
           'conv_kernel<128,64,2,2>', 'conv_kernel<128,32,4,1>',
           'wgrad_kernel<64,192,1,4>', 'wgrad_kernel<192,128,4,2>', 'wgrad_kernel<128,128,2,4>',
           'upce_kernel<19>', 'wgrad_kernel<48,192,1,4>', 'wgrad_kernel<32,192,1,4>', 'dgrad_group_kernel', 'conv3x3_halo_kernel',
           'conv1x1_stream_kernel<192>', 'conv1x1_stream_kernel<128>', 'wgrad_group_kernel', 'dgrad3x3_fused_kernel', 'dgrad_wide_kernel', 'conv_wide_kernel<256>',
           'wgrad_wide_kernel', 'wgrad3x3_strip_kernel', 'conv_wide_kernel<192>', 'conv_wide_kernel<128>', 'stem7_fwd_kernel',
           'stem7_wgrad_kernel']


def kernel_class(kind, f):
    if kind == 'conv':
        co = f['cout']
        if f.get('gx_corr1') is not None:
            return 17
        if (f.get('g') is not None and (f.get('n_src') or 0) <= 1 and f.get('g_bf16') and f['taps'] == 1 and f.get('epi_relu') == 1
                and f.get('ref') is not None and f.get('e_scale') is not None and f.get('e_mean') is not None and f.get('in_scale') is None
                and f.get('in_mask') is None and not f.get('in_relu') and f.get('bias') is None and f.get('out_mask') is None
                and not f.get('dtype') and f['cin'] >= 256 and co >= 512 and co % 4 == 0 and os.environ.get('DSASPP_DGRAD_WIDE', '1') != '0'):
            return 18
        if (f['taps'] == 9 and f.get('g') is None and f.get('epi_relu') == 1 and co >= 448 and co % 4 == 0 and f['cin'] >= 32 and 9 * f['cin'] >= 256
                and f.get('ref') is not None and f.get('e_scale') is not None and f.get('e_mean') is not None and f.get('in_scale') is None
                and f.get('in_mask') is None and not f.get('in_relu') and f.get('bias') is None and f.get('out_mask') is None
                and not f.get('dtype') and not f.get('y_f32') and f.get('stat_acc') is None and os.environ.get('DSASPP_DGRAD_WIDE', '1') != '0'):
            return 18                          # dilated 3x3 data gradient with many output channels (ASPP) on the wide tile
        if (f.get('n_src') or 0) > 1 or (
                f.get('g') is not None and co >= int(os.environ.get('DSASPP_DGG_MIN_COUT', 48)) and f['taps'] == 1 and f.get('epi_relu') and f.get('ref') is not None
                and f.get('e_scale') is not None and f.get('e_mean') is not None and f.get('in_scale') is None
                and f.get('in_mask') is None and not f.get('in_relu') and f.get('bias') is None and f.get('out_mask') is None
                and f.get('epi_relu') == 1 and f['cin'] % 8 == 0 and 64 <= f['cin'] <= 4096 and f['ldx'] % 8 == 0 and co % 4 == 0):
            return 12
        if (f['taps'] == 9 and 1 <= f['dil'] <= 4 and co <= 64 and f['cin'] % 64 == 0 and f.get('g') is None
                and not f.get('epi_relu') and f.get('s_acc') is None and not f.get('y_f32') and f.get('in_mask') is None
                and f.get('bias') is None and f.get('out_mask') is None):
            return 13
        if (f['taps'] == 1 and 64 < co <= 192 and f.get('in_relu') in (0, 1, None) and f.get('in_mask') is None
                and f.get('out_mask') is None and not f.get('y_f32') and not f.get('epi_relu') and f.get('s_acc') is None
                and f.get('g') is None and co % 4 == 0 and not f.get('dtype') and (f.get('n_src') or 0) <= 1
                and os.environ.get('DSASPP_CONV_WIDE', '1') != '0'):
            # BN-ReLU-1x1 forward on the 256 x 192 / 256 x 128 tile (conv_wide.hip), in the K range where it is used
            env = os.environ.get
            if (co > 128 and int(env('DSASPP_CONV_WIDE192_MIN_CIN', 64)) <= f['cin'] <= int(env('DSASPP_CONV_WIDE192_MAX_CIN', 1392))
                    and env('DSASPP_CONV_WIDE192', '1') != '0'):
                return 22
            if co <= 128 and 64 <= f['cin'] <= int(env('DSASPP_CONV_WIDE128_MAX_CIN', 512)) and env('DSASPP_CONV_WIDE128', '1') != '0':
                return 23
        if (f['taps'] == 1 and 96 <= co <= 192 and f.get('in_scale') is not None and f.get('in_relu') == 1
                and f.get('in_mask') is None and f.get('bias') is None and f.get('out_mask') is None and not f.get('y_f32')
                and not f.get('epi_relu') and f.get('s_acc') is None and f.get('g') is None and f['cin'] >= 64
                and not f.get('dtype')):
            return 14 if co > 128 else 15      # streaming BN-ReLU-1x1 forward (conv1x1_stream.hip)
        if (f['taps'] == 1 and f.get('in_relu') in (0, 1, None) and f.get('in_mask') is None
                and f.get('out_mask') is None and not f.get('y_f32') and not f.get('epi_relu') and f.get('s_acc') is None
                and f.get('g') is None and f['cin'] >= 512 and co >= 448 and co % 4 == 0 and not f.get('dtype')
                and (f.get('n_src') or 0) <= 1 and os.environ.get('DSASPP_CONV_WIDE', '1') != '0'):
            return 19
        if co <= 32:
            return 5
        if co <= 48:
            return 3
        if co <= 64:
            return 4
        pad192, pad128 = (co + 191) // 192 * 192, (co + 127) // 128 * 128
        return 1 if pad192 <= pad128 else 2
    if kind == 'wgrad':
        co = f['cout']
        if (f.get('n_src') or 0) > 1:
            return 16
        if (f['taps'] == 1 and f.get('in_mask') is None and f.get('in_relu') in (0, 1, None) and f['cin'] >= 512 and co >= 448
                and not f.get('dtype') and os.environ.get('DSASPP_WGRAD_WIDE', '1') != '0'):
            return 20
        if (f['taps'] == 9 and 1 <= f['dil'] <= 4 and f['W'] % 32 == 0 and f['cin'] % 64 == 0 and co in (32, 48)
                and f.get('in_mask') is None and f.get('in_relu') in (0, 1, None) and not f.get('dtype')
                and os.environ.get('DSASPP_WGRAD3_STRIP', '1') != '0'):
            return 21
        if (f['taps'] == 9 and f['W'] >= 32 and f.get('in_mask') is None and f.get('in_relu') in (0, 1, None) and f['cin'] >= 448
                and 9 * co >= 448 and not f.get('dtype') and os.environ.get('DSASPP_WGRAD_WIDE', '1') != '0'):
            return 20                          # dilated 3x3 with many input channels (ASPP): the wide tile over nine shifted copies of dz
        if co <= 32:
            return 11
        if co <= 48:
            return 10
        if co <= 64:
            return 6
        if co > 128 and (co + 191) // 192 * 192 <= (co + 127) // 128 * 128:
            return 7
        return 8
    if kind == 'upce':
        return 9
    if kind == 'stem7_fwd':
        return 24
    if kind == 'stem7_wgrad':
        return 25
    return 0


def work(kind, f):
    """(flops, bytes) of one launch: algorithmic, i.e. what the operation needs when every operand is moved once (each
    input tensor read once, each output written once, a gradient slab that is accumulated into read and written),
    not what a kernel moves."""
    if kind == 'wgrad':
        M = f['n_img'] * f['H'] * f['W']
        n = max(1, f.get('n_src') or 0)
        cins = [f['cin']] + [f['cin%d' % i] for i in range(1, n)]
        k = sum(cins) * f['taps']
        # the shared input once, one dz per source, fp32 weight gradients out
        return 2.0 * M * f['cout'] * k, 2.0 * M * max(cins) + 2.0 * M * f['cout'] * n + 4.0 * f['cout'] * k
    if kind == 'conv':
        M = f['n_img'] * f['H'] * f['W']
        n = max(1, f.get('n_src') or 0)
        esz = 4.0 if f.get('dtype') else 2.0
        by = esz * M * f['cin'] * n + esz * f['cout'] * f['taps'] * f['cin'] * n
        if f.get('ref') is not None:
            by += 2.0 * M * f['cout']
        if f.get('gx_corr1') is not None:      # the activation slice of the read-out, the read-out copy written
            by += 2.0 * M * f['cin'] * (2 if f.get('gx_dz') is not None else 1)
        if f.get('g') is not None:
            gsz = 2.0 if f.get('g_bf16') else 4.0
            by += gsz * M * f['cout'] * (2 if f.get('g_accumulate') else 1)
        else:
            by += (4.0 if f.get('y_f32') else esz) * M * f['cout']
        return 2.0 * M * f['cout'] * f['taps'] * f['cin'] * n, by
    if kind in ('stem7_fwd', 'stem7_wgrad'):
        M0 = f['n_img'] * (f['H'] // 2) * (f['W'] // 2)
        img = 4.0 * 3 * f['n_img'] * f['H'] * f['W']
        # forward: image in, bf16 output out; weight gradient: image, masked gradient and forward output in
        by = img + 2.0 * M0 * f['cout'] * (1 if kind == 'stem7_fwd' else 2)
        return 2.0 * M0 * f['cout'] * 147, by
    if kind == 'upce':
        m = f['n_img'] * f['h'] * f['w']
        # fp32 low-res logits read (n_class), int64 labels read, fp32 low-res gradient written
        return 0.0, m * f['n_class'] * 4.0 + m * 64 * 8.0 + m * f['n_class'] * 4.0
    return 0.0, 0.0


def role(kind, f):
    """Finer grouping used for reporting only: what the launch does in the network."""
    if kind == 'conv':
        if f.get('g') is not None:
            return 'dgrad1x1->slab x%d' % f['n_src'] if (f.get('n_src') or 0) > 1 else 'dgrad1x1->slab'
        if f.get('epi_relu'):
            return 'dgrad3x3' if f['taps'] == 9 else 'dgrad1x1'
        return 'fwd3x3' if f['taps'] == 9 else 'fwd1x1'
    if kind == 'wgrad':
        if (f.get('n_src') or 0) > 1:
            return 'wgrad1x1 x%d' % f['n_src']
        return 'wgrad3x3' if f['taps'] == 9 else 'wgrad1x1'
    return kind


def tag_program_fine(prog, table):
    """Tag ops by (kernel class, role); `table` maps the key to a tag and is extended in place."""
    per = {}
    for i, (kind, f) in enumerate(prog.keepalive):
        t = kernel_class(kind, f)
        if not t:
            continue
        key = CLASSES[t] + ' ' + role(kind, f)
        tag = table.setdefault(key, len(table) + 1)
        prog.set_tag(i, tag)
        fl, by = work(kind, f)
        a = per.setdefault(tag, [0.0, 0.0, 0])
        a[0] += fl
        a[1] += by
        a[2] += 1
    return per


def tag_program(prog, only=None):
    """Tag every MFMA / loss op of a HipProgram (or those of kernel class `only`) for event timing; returns per-class
    (flops, bytes, launches) per run."""
    per = {}
    for i, (kind, f) in enumerate(prog.keepalive):
        t = kernel_class(kind, f)
        if t and (only is None or t == only):
            prog.set_tag(i, t)
            fl, by = work(kind, f)
            a = per.setdefault(t, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += by
            a[2] += 1
    return per


def untag_program(prog):
    for i, (kind, f) in enumerate(prog.keepalive):
        if kernel_class(kind, f):
            prog.set_tag(i, 0)
