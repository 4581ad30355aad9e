"""ctypes binding of libmmdyn_hip.so (include/mmdyn_hip.h).

The HIP library IS the product's compute path: there is no CPU or PyTorch fallback.  If the shared
object is missing or a symbol is absent the import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMDYN_HIP_LIB") or os.path.join(_HERE, "libmmdyn_hip.so")   # override: kernel experiments

_P, _I, _L, _F, _Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64


class PassExperts(ctypes.Structure):
    """mmdyn_pass_experts (include/mmdyn_hip.h)."""
    _fields_ = [("mu", _P * 4), ("lv", _P * 4), ("dmu", _P * 4), ("dlv", _P * 4), ("ld", _I * 4), ("dz", _P * 3),
                ("zdst", _P * 3), ("zpl", _P * 3)]


class PackEntry(ctypes.Structure):
    """mmdyn_pack_entry (include/mmdyn_hip.h)."""
    _fields_ = [("src", _P), ("dst", _P), ("kind", _I), ("rows_in", _I), ("cols_in", _I), ("rows_out", _I),
                ("cols_out", _I), ("ld_out", _I), ("dst_bf16", _I)]


MAX_PASSES = 8
MAX_EXPERTS = 4
ABI_VERSION = 6          # MMDYN_ABI_VERSION of the include/mmdyn_hip.h this table was written against

# name -> argument type codes, in header order: p pointer, i int, l int64, f float, Q uint64
_SIGNATURES = {
    "mmdyn_abi_version": "",
    "mmdyn_igemm_nt": "ppppppp" + "iiiiiiiiiiiiii" + "p",
    "mmdyn_igemm_nt_dgrad_bn": "ppppppppp" + "iiiiiiiiiiii" + "ppp",
    "mmdyn_igemm_nt_dgrad_act": "pppp" + "i" + "iiiiiiiiiii" + "i" + "ppp",
    "mmdyn_igemm_slab_floats": "iiiiiiiii",
    "mmdyn_igemm_nt_bf16": "ppppppp" + "iiiiiiiiiiiiii" + "p",
    "mmdyn_igemm_nt_f16": "ppppppp" + "iiiiiiiiiiiiii" + "p",
    "mmdyn_igemm_stat_tiles": "iiiiiiiii",
    "mmdyn_igemm_stat_tiles_bf16": "iiiiiiiii",
    "mmdyn_igemm_stat_tiles_mx": "iiiiiiiiii",
    "mmdyn_igemm_slab_floats_mx": "iiiiiiiiii",
    "mmdyn_igemm_nt_grouped": "pppppp" + "iiiiii" + "p",
    "mmdyn_splitk_reduce": "pppp" + "iiii" + "p",
    "mmdyn_wgrad_tn": "ppp" + "iiiiiiiiiii" + "p",
    "mmdyn_wgrad_tn_bf16": "ppp" + "iiiiiiiiiii" + "p",
    "mmdyn_wgrad_tn_f16": "ppp" + "iiiiiiiiiii" + "p",
    "mmdyn_wgrad_tn_grouped": "ppp" + "iiiiii" + "p",
    "mmdyn_wgrad_chunks": "iiii",
    "mmdyn_wgrad_chunks_mx": "iiiii",
    "mmdyn_wgrad_reduce": "pp" + "iiiiii" + "f" + "p",
    "mmdyn_pack_conv_weight": "pp" + "iii" + "p",
    "mmdyn_repack2d": "pp" + "iiiii" + "p",
    "mmdyn_repack2d_ld": "pp" + "iiiiii" + "p",
    "mmdyn_pack_conv_weight_b16": "pp" + "iiii" + "p",
    "mmdyn_repack2d_ld_b16": "pp" + "iiiiiii" + "p",
    "mmdyn_pack_plan": "p" + "i" + "p",
    "mmdyn_im2col_nchw3": "pp" + "iii" + "p",
    "mmdyn_col2im_k4": "pp" + "iiiiiiiiii" + "p",
    "mmdyn_tconv_out3_fwd": "ppp" + "iii" + "p",
    "mmdyn_tconv_out3_bn_fwd": "ppppppp" + "iiiii" + "p",
    "mmdyn_tconv_out3_bn_bce": "ppppppp" + "i" + "pp" + "i" + "pppp" + "f" + "iiiii" + "p",
    "mmdyn_wgrad_out3_bn": "ppppppp" + "iiiii" + "p",
    "mmdyn_colstats": "pp" + "iii" + "p",
    "mmdyn_colstats_tiles": "i",
    "mmdyn_bn_finalize": "ppppppp" + "iiii" + "ff" + "i" + "pp",
    "mmdyn_bn_swish_fwd": "pppppp" + "iii" + "p",
    "mmdyn_bn_swish_bwd_reduce": "ppppppp" + "iii" + "p",
    "mmdyn_bn_bwd_finalize": "ppppp" + "iii" + "f" + "pp",
    "mmdyn_bn_eval_stats": "pppp" + "ii" + "f" + "p",
    "mmdyn_bn_reduce_partials": "ppp" + "iii" + "p",
    "mmdyn_bn_finalize_sums": "pppppp" + "iii" + "ff" + "i" + "p",
    "mmdyn_bn_bwd_finalize_sums": "pppp" + "ii" + "ff" + "p",
    "mmdyn_bn_swish_bwd_apply": "pppppppp" + "iiii" + "p",
    "mmdyn_act_fwd": "pp" + "l" + "i" + "p",
    "mmdyn_act_bwd": "ppp" + "l" + "i" + "p",
    "mmdyn_dropout_expand": "ppp" + "iii" + "f" + "p",
    "mmdyn_dropout_reduce": "ppp" + "iii" + "f" + "pi" + "pp",
    "mmdyn_random_masks": "p" + "l" + "f" + "QQ" + "pp",
    "mmdyn_random_normal": "p" + "l" + "QQ" + "pp",
    "mmdyn_counter_add": "p" + "Q" + "p",
    "mmdyn_colsum": "ppp" + "iii" + "f" + "pp",
    "mmdyn_colsum_chunks": "i",
    "mmdyn_scale_dev": "ppp" + "l" + "p",
    "mmdyn_sum_blocks": "pp" + "i" + "l" + "p",
    "mmdyn_copy_many": "ppp" + "i" + "p",
    "mmdyn_cast_f32_to_bf16": "pp" + "l" + "p",
    "mmdyn_cast_bf16_to_f32": "pp" + "l" + "p",
    "mmdyn_linear_small_fwd": "pppp" + "iiii" + "p",
    "mmdyn_linear_small_bwd": "pppppp" + "iii" + "f" + "p",
    "mmdyn_poe_fwd": "pppppp" + "iiii" + "p",
    "mmdyn_poe_bwd": "ppppppp" + "f" + "iiii" + "pp",
    "mmdyn_reparam_fwd": "ppppp" + "iii" + "p",
    "mmdyn_reparam_bwd": "pppp" + "f" + "pp" + "iii" + "p",
    "mmdyn_bce_logits": "ppppp" + "l" + "iii" + "f" + "p",
    "mmdyn_bce_logits_groups": "ppppp" + "i" + "l" + "f" + "p",
    "mmdyn_bce_logits_groups_masked": "ppppppp" + "i" + "l" + "iii" + "f" + "p",
    "mmdyn_mse": "pppp" + "l" + "f" + "p",
    "mmdyn_mse_groups": "ppppp" + "i" + "l" + "f" + "p",
    "mmdyn_elbo_assemble": "ppppp" + "ii" + "ff" + "pp",
    "mmdyn_adam_step": "ppppp" + "l" + "fffff" + "p",
    "mmdyn_adam_step_guarded": "ppppp" + "l" + "fffff" + "p",
    "mmdyn_sgd_step": "ppp" + "l" + "ffff" + "i" + "p",
    "mmdyn_igemm_nt_mx": "pppppppppppp" + "iiiiiiiiiiiiii" + "i" + "pp",
    "mmdyn_wgrad_tn_mx": "ppp" + "iiiiiiiiiii" + "i" + "p",
    "mmdyn_split_planes": "pp" + "l" + "i" + "p",
    "mmdyn_igemm_planes_served": "iiiiiiiii",
    "mmdyn_bn_swish_fwd_planes": "ppppppp" + "iii" + "p",
    "mmdyn_bn_swish_bwd_apply_planes": "ppppppppp" + "iiii" + "p",
    "mmdyn_bn_swish_fwd_b16": "pppppp" + "iiii" + "p",
    "mmdyn_bn_swish_bwd_reduce_b16": "ppppppp" + "iiii" + "p",
    "mmdyn_bn_swish_bwd_apply_b16": "pppppppp" + "iiiii" + "p",
    "mmdyn_act_bwd_b16": "ppp" + "l" + "ii" + "p",
    "mmdyn_tconv_out3_fwd_b16": "ppp" + "iiii" + "p",
    "mmdyn_resize_ksize": "ii",
    "mmdyn_resize_plan": "ii" + "pp",
    "mmdyn_resize_u8_to_chw_f32": "ppp" + "iiiii" + "pppp" + "p",
    "mmdyn_nchw_to_nhwc": "pp" + "iii" + "p",
    "mmdyn_nhwc_to_nchw": "pp" + "iii" + "p",
}
_CODES = {"p": _P, "i": _I, "l": _L, "f": _F, "Q": _Q}

EXPORTS = ["mmdyn_version"] + list(_SIGNATURES)

# the LAB build of the same sources (make lab: -DMMDYN_LAB): experiment environment variables + the opt-in direct-fragment
# kernels.  Never loaded by the product; tests/microbench and the variant-forcing kernel tests ask for it by path.
LAB_LIB_PATH = os.path.join(_HERE, "libmmdyn_hip_lab.so")

_libs = {}


def load(path=None):
    """Load the library (once per path); raises OSError / AttributeError if it (or a symbol) is missing."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: build it with `make -C multimodal-dynamics_amd/csrc`"
            f"{' lab' if path == LAB_LIB_PATH else ''} "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no fallback path.")
    lib = ctypes.CDLL(path)
    lib.mmdyn_version.restype = ctypes.c_char_p
    lib.mmdyn_version.argtypes = []
    for name, sig in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = [_CODES[c] for c in sig]
    got = lib.mmdyn_abi_version()
    if got != ABI_VERSION:
        raise OSError(f"{path} implements ABI revision {got}, this binding was written against {ABI_VERSION}: rebuild the "
                      "library (make -C multimodal-dynamics_amd/csrc) -- signatures and workspace arguments differ between revisions")
    _libs[path] = lib
    return lib


class MmdynError(RuntimeError):
    pass


_ERR = {-1: "MMDYN_ERR_SHAPE (unsupported dimensions)", -2: "MMDYN_ERR_NULL (null pointer)",
        -3: "MMDYN_ERR_RANGE (tensor too large for 32-bit offsets)"}


def check(rc, name):
    if rc != 0:
        raise MmdynError(f"{name} failed: {_ERR.get(rc, 'hipError_t %d' % rc)}")
