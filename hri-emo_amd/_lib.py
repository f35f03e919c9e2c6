"""ctypes binding of libhriemo.so (include/hriemo.h).  There is NO fallback: if the library is missing
or a kernel call fails this raises -- the product path never routes around the HIP kernels."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HRIEMO_LIB: developer override (scripts_dev/forensics builds variant libraries); a path that does not exist still raises
LIB_PATH = os.environ.get("HRIEMO_LIB") or os.path.join(_HERE, "libhriemo.so")

_C = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float,
      "Q": ctypes.c_ulonglong, "I": ctypes.c_uint}

# name -> (argument codes, result): p pointer, i int, l long, f float, Q u64, I u32
_SIGS = {
    "hriemo_gemm_bf16": ("iiiiiplplplipipliplp", "i"),
    "hriemo_gemm_force_config": ("i", "i"),
    "hriemo_gemm_debug_flags": ("i", "i"),
    "hriemo_gemm_colsum_rows": ("iiiii", "i"),
    "hriemo_gemm_bf16_group_tn": ("piip", "i"),
    "hriemo_gemm_bf16_split": ("iiiiiplplplpliiplp", "i"),
    "hriemo_gemm_bf16_colsum": ("iiiiiplplplplpp", "i"),
    "hriemo_mx8_scale_ld": ("i", "l"),
    "hriemo_quant_mx8": ("pliiiplplp", "i"),
    "hriemo_gemm_mx8": ("iiiplplplplplipiplp", "i"),
    "hriemo_gemm_mx8_q": ("iiiplplplplplpiplplp", "i"),
    "hriemo_gemm_mx8_force_config": ("i", "i"),
    "hriemo_attn_fwd": ("plplplplppiiiiifQpIipp", "i"),
    "hriemo_attn_fwd_q": ("plplplplppiiiiifQpIipplplp", "i"),
    "hriemo_attn_bwd": ("plplplplplplplplpppiiiiifQpIipppp", "i"),
    "hriemo_attn_fwd_varlen": ("plplplplpppiiiiifQpIipp", "i"),
    "hriemo_attn_bwd_varlen": ("plplplplplplplplppppiiiiifQpIipppp", "i"),
    "hriemo_attn_mask_bytes": ("iiii", "l"),
    "hriemo_attn_bwd_single_pass": ("iiii", "i"),
    "hriemo_attn_bwd_single_pass_q": ("iiiii", "i"),
    "hriemo_attn_bwd_kv_colsum_rows": ("iiiii", "i"),
    "hriemo_attn_bwd_colsum_rows": ("iiii", "i"),
    "hriemo_attn_bwd_dq_colsum_rows": ("iiiii", "i"),
    "hriemo_split_bf16x3": ("pliipiip", "i"),
    "hriemo_attn_fwd_f32": ("plplplplppiiiiifQpIip", "i"),
    "hriemo_attn_probs_f32": ("plplpppiiiiifQpIip", "i"),
    "hriemo_add_ln_f32": ("ppppppiiffQpIlp", "i"),
    "hriemo_dropout_f32": ("ppliipfQpIlp", "i"),
    "hriemo_masked_mean_f32": ("pppiiip", "i"),
    "hriemo_gate_input_f32": ("pppiip", "i"),
    "hriemo_sigmoid_beta_f32": ("pppiip", "i"),
    "hriemo_fuse_f32": ("ppipippiiip", "i"),
    "hriemo_split3_f32": ("pliipiiplp", "i"),
    "hriemo_colsum_f32_workspace_bytes": ("ii", "l"),
    "hriemo_colsum_f32": ("pliiplpipp", "i"),
    "hriemo_add_ln_bwd_f32_workspace_bytes": ("ii", "l"),
    "hriemo_add_ln_bwd_f32": ("pppppppppiiiffQpIlpp", "i"),
    "hriemo_attn_bwd_f32": ("plplplplplppplplplpiiiiifQpIip", "i"),
    "hriemo_gate_dpre_f32": ("ppipipppiiip", "i"),
    "hriemo_gate_input_bwd_f32": ("pppppiip", "i"),
    "hriemo_gate_dy_f32": ("ppipppiiiip", "i"),
    "hriemo_rowdot_bwd_f32": ("ppppppiiip", "i"),
    "hriemo_attn_probs": ("plplpppiiiiifQpIip", "i"),
    "hriemo_add_ln_fwd": ("pppppppppiiffQpIlp", "i"),
    "hriemo_add_ln_fwd_mx8": ("pppppppppiiffQpIlpplp", "i"),
    "hriemo_add_ln_fwd_rows": ("pppppppppiiffQpIlpplpp", "i"),
    "hriemo_add_ln_bwd_rows": ("ppppppppppppiiifQpIlppp", "i"),
    "hriemo_add_ln_bwd_workspace_bytes": ("ii", "l"),
    "hriemo_add_ln_bwd": ("ppppppppppppiiifQpIlpp", "i"),
    "hriemo_colsum_workspace_bytes": ("ii", "l"),
    "hriemo_colsum_bf16": ("pliipipp", "i"),
    "hriemo_cast_f32_to_bf16": ("pplp", "i"),
    "hriemo_cast_bf16_to_f32": ("pplp", "i"),
    "hriemo_cast_f32_to_bf16_batch": ("pip", "i"),
    "hriemo_cast_copy_batch": ("pip", "i"),
    "hriemo_pack_rows": ("pppiiiipppp", "i"),
    "hriemo_unpack_rows": ("pppiiippp", "i"),
    "hriemo_dropout_bf16": ("pplifQpIlp", "i"),
    "hriemo_gemm_ln_supported": ("i", "i"),
    "hriemo_gemm_ln_fwd": ("iiiplplppppppppppffQpIlpp", "i"),
    "hriemo_expand_rows": ("pppilp", "i"),
    "hriemo_seed_bump": ("pp", "i"),
    "hriemo_rowdot_fwd": ("pppppiip", "i"),
    "hriemo_rowdot_bwd": ("pppppppiiip", "i"),
    "hriemo_pool_chunks": ("i", "i"),
    "hriemo_ln_pool_fwd": ("pppppppppiiiifp", "i"),
    "hriemo_gate_input": ("ppppiiiipppp" + "p", "i"),
    "hriemo_sigmoid_beta": ("pppiip", "i"),
    "hriemo_fuse_fwd": ("ppppiiip", "i"),
    "hriemo_add_ln_bwd_partial_rows": ("ii", "i"),
    "hriemo_rowops_force_variant": ("i", "i"),
    "hriemo_colsum_partial_rows": ("ii", "i"),
    "hriemo_colreduce_batch": ("pipip", "i"),
    "hriemo_debug_hog": ("iipp", "i"),
    "hriemo_sumsq_f32": ("plpip", "i"),
    "hriemo_adamw_flat": ("pppplfffffifpp", "i"),
    "hriemo_masked_mean_fwd": ("ppppiiip", "i"),
    "hriemo_rowsum_f32": ("ppilp", "i"),
    "hriemo_gate_input_pooled": ("pppiip", "i"),
    "hriemo_fusion_loss": ("ppppiiiffpppp", "i"),
    "hriemo_fusion_loss_ce": ("pppiiiffpppp", "i"),
    "hriemo_scalar_gate_dx": ("pipippppiiip", "i"),
    "hriemo_fuse_bwd_dw": ("ppppiiip", "i"),
    "hriemo_gate_dpre": ("pippp" + "iip", "i"),
    "hriemo_gate_input_bwd": ("ppppppiip", "i"),
    "hriemo_ln_pool_bwd_chunks": ("i", "i"),
    "hriemo_ln_pool_pair_supported": ("i", "i"),
    "hriemo_ln_pool_fwd_pair": ("pppppppppi" * 2 + "iiifp", "i"),
    "hriemo_ln_pool_bwd_pair": ("pip" + "ppppppppppip" * 2 + "iiip", "i"),
    "hriemo_ln_pool_bwd_workspace_bytes": ("iii", "l"),
    "hriemo_ln_pool_bwd": ("pipipppppppppp" + "iiiipp", "i"),
    "hriemo_prof_enable": ("i", "i"),
    "hriemo_prof_nclass": ("", "i"),
    "hriemo_prof_collect": ("ippp", "i"),
    "hriemo_abi_version": ("", "i"),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C hri-emo_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU/eager fallback for this path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = [_C[c] for c in args]
            fn.restype = _C[res]
        L.hriemo_last_error.restype = ctypes.c_char_p
        L.hriemo_last_error.argtypes = []
        L.hriemo_prof_name.restype = ctypes.c_char_p
        L.hriemo_prof_name.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def call(name, *args):
    """Invoke an int-status entry point; raise with the library's message on failure."""
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.hriemo_last_error().decode()}")
