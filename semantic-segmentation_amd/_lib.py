"""ctypes binding of libdsaspp.so.

The descriptor structs are generated from include/dsaspp.h at import time, so the Python
view of every struct is the header's by construction; dsaspp_program_add additionally
checks each descriptor's size against the library's own sizeof.

The library is the only compute path: importing this module without a built
libdsaspp.so raises, there is no fallback.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'dsaspp.h')
# DSASPP_LIB_PATH: measure another build of the same library (kernel A/B experiments); default is the in-tree build
LIB_PATH = os.environ.get('DSASPP_LIB_PATH') or os.path.join(HERE, 'libdsaspp.so')

_SCALARS = {
    'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
    'uint8_t': ctypes.c_uint8, 'int': ctypes.c_int,
}


def _parse_header(path):
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    structs = {}
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            is_ptr = '*' in decl
            decl_clean = decl.replace('const', ' ').replace('*', ' ')
            parts = decl_clean.split()
            ctype_name = parts[0]
            names = ' '.join(parts[1:]).split(',')
            for n in names:
                n = n.strip()
                if not n:
                    continue
                if is_ptr:
                    fields.append((n, ctypes.c_void_p))
                else:
                    fields.append((n, _SCALARS[ctype_name]))
        structs[name] = type(name, (ctypes.Structure,), {'_fields_': fields})
    enums = {}
    for m in re.finditer(r'enum\s*\{(.*?)\}\s*;', text, flags=re.S):
        val = 0
        for item in m.group(1).split(','):
            item = item.strip()
            if not item:
                continue
            if '=' in item:
                k, v = item.split('=')
                val = int(v.strip(), 0)
                enums[k.strip()] = val
            else:
                enums[item] = val
            val += 1
    defines = {k: int(v, 0) for k, v in re.findall(r'#define\s+(DSASPP_\w+)\s+(\d+)', text)}
    return structs, enums, defines


STRUCTS, ENUMS, DEFINES = _parse_header(HEADER)
STAT_REPLICAS = DEFINES['DSASPP_STAT_REPLICAS']

conv_desc = STRUCTS['dsaspp_conv_desc']
wgrad_desc = STRUCTS['dsaspp_wgrad_desc']
bn_desc = STRUCTS['dsaspp_bn_desc']
bn_bwd_desc = STRUCTS['dsaspp_bn_bwd_desc']
bnact_desc = STRUCTS['dsaspp_bnact_desc']
gradx_desc = STRUCTS['dsaspp_gradx_desc']
bnapply_desc = STRUCTS['dsaspp_bnapply_desc']
pool_desc = STRUCTS['dsaspp_pool_desc']
im2col_desc = STRUCTS['dsaspp_im2col_desc']
maxpool_desc = STRUCTS['dsaspp_maxpool_desc']
ce_desc = STRUCTS['dsaspp_ce_desc']
up_desc = STRUCTS['dsaspp_up_desc']
adam_desc = STRUCTS['dsaspp_adam_desc']
wt_entry = STRUCTS['dsaspp_wt_entry']
generic_desc = STRUCTS['dsaspp_generic_desc']
stem3_desc = STRUCTS['dsaspp_stem3_desc']
dw_desc = STRUCTS['dsaspp_dw_desc']
conf_desc = STRUCTS['dsaspp_conf_desc']
prep_desc = STRUCTS['dsaspp_prep_desc']
up2_desc = STRUCTS['dsaspp_up2_desc']
dwbwd_desc = STRUCTS['dsaspp_dwbwd_desc']
col2im_desc = STRUCTS['dsaspp_col2im_desc']
stem7_desc = STRUCTS['dsaspp_stem7_desc']

# op name -> (program op code, descriptor struct, direct entry point)
OPS = {
    'conv': (ENUMS['DSASPP_OP_CONV'], conv_desc, 'dsaspp_conv_fwd'),
    'wgrad': (ENUMS['DSASPP_OP_WGRAD'], wgrad_desc, 'dsaspp_conv_wgrad'),
    'bn_finalize': (ENUMS['DSASPP_OP_BN_FINALIZE'], bn_desc, 'dsaspp_bn_finalize'),
    'bn_bwd_finalize': (ENUMS['DSASPP_OP_BN_BWD_FINALIZE'], bn_bwd_desc, 'dsaspp_bn_bwd_finalize'),
    'bn_act': (ENUMS['DSASPP_OP_BN_ACT'], bnact_desc, 'dsaspp_bn_act'),
    'grad_extract': (ENUMS['DSASPP_OP_GRAD_EXTRACT'], gradx_desc, 'dsaspp_grad_extract'),
    'bn_bwd_apply': (ENUMS['DSASPP_OP_BN_BWD_APPLY'], bnapply_desc, 'dsaspp_bn_bwd_apply'),
    'avgpool_fwd': (ENUMS['DSASPP_OP_AVGPOOL_FWD'], pool_desc, 'dsaspp_avgpool2_fwd'),
    'avgpool_bwd': (ENUMS['DSASPP_OP_AVGPOOL_BWD'], pool_desc, 'dsaspp_avgpool2_bwd'),
    'im2col': (ENUMS['DSASPP_OP_IM2COL'], im2col_desc, 'dsaspp_stem_im2col'),
    'maxpool_fwd': (ENUMS['DSASPP_OP_MAXPOOL_FWD'], maxpool_desc, 'dsaspp_maxpool3s2_fwd'),
    'maxpool_bwd': (ENUMS['DSASPP_OP_MAXPOOL_BWD'], maxpool_desc, 'dsaspp_maxpool3s2_bwd'),
    'upce': (ENUMS['DSASPP_OP_UPCE'], ce_desc, 'dsaspp_upsample8_ce'),
    'ce_finalize': (ENUMS['DSASPP_OP_CE_FINALIZE'], ce_desc, 'dsaspp_ce_finalize'),
    'up_fwd': (ENUMS['DSASPP_OP_UP_FWD'], up_desc, 'dsaspp_upsample8_fwd'),
    'up_bwd': (ENUMS['DSASPP_OP_UP_BWD'], up_desc, 'dsaspp_upsample8_bwd'),
    'up_argmax': (ENUMS['DSASPP_OP_UP_ARGMAX'], up_desc, 'dsaspp_upsample8_argmax'),
    'memset': (ENUMS['DSASPP_OP_MEMSET'], generic_desc, None),
    'copy2d': (ENUMS['DSASPP_OP_COPY2D'], generic_desc, None),
    'copy2d_bf16': (ENUMS['DSASPP_OP_COPY2D_BF16'], generic_desc, None),
    'stem3': (ENUMS['DSASPP_OP_STEM3'], stem3_desc, 'dsaspp_stem3x3s2'),
    'dwconv': (ENUMS['DSASPP_OP_DWCONV'], dw_desc, 'dsaspp_dwconv3x3'),
    'up2_fwd': (ENUMS['DSASPP_OP_UP2_FWD'], up2_desc, 'dsaspp_upsample2_fwd'),
    'up2_bwd': (ENUMS['DSASPP_OP_UP2_BWD'], up2_desc, 'dsaspp_upsample2_bwd'),
    'dw_bwd_data': (ENUMS['DSASPP_OP_DW_BWD_DATA'], dwbwd_desc, 'dsaspp_dwconv3x3_bwd_data'),
    'dw_wgrad': (ENUMS['DSASPP_OP_DW_WGRAD'], dwbwd_desc, 'dsaspp_dwconv3x3_wgrad'),
    'stem3_col': (ENUMS['DSASPP_OP_STEM3_COL'], im2col_desc, 'dsaspp_stem3_im2col'),
    'col2im': (ENUMS['DSASPP_OP_COL2IM'], col2im_desc, 'dsaspp_stem_col2im'),
    'stem7_fwd': (ENUMS['DSASPP_OP_STEM7_FWD'], stem7_desc, 'dsaspp_stem7_fwd'),
    'stem7_wgrad': (ENUMS['DSASPP_OP_STEM7_WGRAD'], stem7_desc, 'dsaspp_stem7_wgrad'),
    # direct calls only (not recordable in a program)
    'confusion': (None, conf_desc, 'dsaspp_upsample8_confusion'),
    'preprocess': (None, prep_desc, 'dsaspp_preprocess'),
}

_lib = None


class DsasppError(RuntimeError):
    pass


def load():
    """Load libdsaspp.so (built in-tree by build.py).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsasppError('libdsaspp.so not found at %s -- build it with '
                          '`python semantic-segmentation_amd/build.py` (needs hipcc); '
                          'there is no fallback compute path' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.dsaspp_last_error.restype = ctypes.c_char_p
    lib.dsaspp_version.restype = ctypes.c_int
    lib.dsaspp_source_hash.restype = ctypes.c_char_p
    lib.dsaspp_conv_last_route.restype = ctypes.c_char_p
    for _, (_, struct, entry) in OPS.items():
        if entry is not None:
            fn = getattr(lib, entry)
            fn.argtypes = [ctypes.POINTER(struct), ctypes.c_void_p]
            fn.restype = ctypes.c_int
    lib.dsaspp_upsample8_ce_loss_noalloc.argtypes = [ctypes.POINTER(ce_desc), ctypes.c_void_p]
    lib.dsaspp_upsample8_ce_loss_noalloc.restype = ctypes.c_int
    lib.dsaspp_adam_step.argtypes = [ctypes.POINTER(adam_desc), ctypes.c_void_p]
    lib.dsaspp_cast_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.dsaspp_weight_transpose.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_int64, ctypes.c_void_p]
    lib.dsaspp_weight_transpose_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                ctypes.c_int64, ctypes.c_void_p]
    lib.dsaspp_copy2d_f32.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.dsaspp_copy2d_f32_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.dsaspp_stem7_wgrad_workspace_floats.argtypes = [ctypes.c_int32]
    lib.dsaspp_stem7_wgrad_workspace_floats.restype = ctypes.c_int64
    lib.dsaspp_set_comm_cus.argtypes = [ctypes.c_int32]
    lib.dsaspp_set_comm_cus.restype = ctypes.c_int
    lib.dsaspp_get_comm_cus.restype = ctypes.c_int
    lib.dsaspp_memset_async.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    lib.dsaspp_program_create.restype = ctypes.c_void_p
    lib.dsaspp_program_destroy.argtypes = [ctypes.c_void_p]
    lib.dsaspp_program_destroy.restype = None
    lib.dsaspp_program_add.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]
    lib.dsaspp_program_size.argtypes = [ctypes.c_void_p]
    lib.dsaspp_program_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.dsaspp_program_set_tag.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.dsaspp_program_set_side.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.dsaspp_program_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_double),
                                               ctypes.POINTER(ctypes.c_int64)]
    _lib = lib
    return lib


def source_hash():
    """Hash of the sources the loaded library was built from (dsaspp_source_hash)."""
    return load().dsaspp_source_hash().decode()


def set_comm_cus(n):
    """Compute units the persistent kernels leave free for a concurrent collective (dsaspp_set_comm_cus)."""
    lib = load()
    if lib.dsaspp_set_comm_cus(int(n)) != 0:
        raise RuntimeError(lib.dsaspp_last_error().decode())


def get_comm_cus():
    return int(load().dsaspp_get_comm_cus())


def conv_last_route():
    """Kernel family the calling thread's most recent convolution or weight-gradient op went to (dsaspp_conv_last_route)."""
    return load().dsaspp_conv_last_route().decode()


def check(rc):
    if rc != 0:
        raise DsasppError(load().dsaspp_last_error().decode('utf-8', 'replace'))


def exported_symbols():
    """Names of every function the header declares (used by the load/ABI test)."""
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dsaspp_\w+)\s*\(', text)))
