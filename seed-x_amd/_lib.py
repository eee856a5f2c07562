"""ctypes binding of libseedx_hip.so (C-ABI in include/seedx_hip.h).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU / eager fallback here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libseedx_hip.so")

SX_F16, SX_BF16, SX_F32 = 0, 1, 2
SX_BF16X3 = 3          # sx_groupnorm* output only: bf16 planes [hi | hi | lo] per row
SX_F16X2 = 4           # sx_groupnorm* output only: fp16 planes [hi | lo] per row
SX_TILED16 = 0x100     # OR-ed into a 16-bit out dtype: decode operand tiles [cols/32][16][32] (include/seedx_hip.h)
SX_ACT_NONE, SX_ACT_GELU, SX_ACT_SILU = 0, 1, 2
SX_A_LINEAR, SX_A_CONV3X3 = 0, 1

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("A", c_vp), ("W", c_vp), ("C", c_vp), ("bias", c_vp), ("bias2d", c_vp), ("residual", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("ldc", c_i32), ("ldr", c_i32), ("n_valid", c_i32),
                ("res_mod", c_i32), ("bias2d_rows", c_i32), ("dtype", c_i32), ("out_dtype", c_i32),
                ("act", c_i32), ("glu", c_i32), ("a_mode", c_i32),
                ("B", c_i32), ("Hin", c_i32), ("Win", c_i32), ("Cin", c_i32), ("Hout", c_i32), ("Wout", c_i32),
                ("stride", c_i32), ("upsample", c_i32), ("ld_bias2d", c_i32), ("pad_mode", c_i32), ("a_planes", c_i32)]


class GemmLnArgs(C.Structure):
    _fields_ = [("x16_out", c_vp), ("row_stats_out", c_vp), ("row_stats_in", c_vp), ("colsum", c_vp),
                ("ld_x16", c_i32), ("dim", c_i32), ("eps", c_f32), ("reserved", c_i32)]


class GemvArgs(C.Structure):
    _fields_ = [("x", c_vp), ("W", c_vp), ("y", c_vp), ("residual", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("dtype", c_i32), ("out_dtype", c_i32), ("act", c_i32), ("glu", c_i32),
                ("w_layout", c_i32), ("x_layout", c_i32), ("workspace", c_vp), ("workspace_bytes", C.c_uint64),
                ("x16_out", c_vp), ("row_ssq_out", c_vp), ("row_ssq_in", c_vp), ("ssq_in_parts", c_i32), ("ssq_dim", c_i32),
                ("ssq_eps", c_f32), ("x_planes", c_i32), ("out_planes", c_i32), ("x16_gamma", c_vp)]


class AttnF32Args(C.Structure):
    _fields_ = [("q", c_vp), ("kcache", c_vp), ("vcache", c_vp), ("out", c_vp), ("pos0_dev", c_vp),
                ("q_row_stride", c_i64), ("cache_seq_stride", c_i64), ("kv_row_stride", c_i64), ("kv_head_stride", c_i64),
                ("G", c_i32), ("T", c_i32), ("H", c_i32), ("D", c_i32), ("Tmax", c_i32), ("dtype", c_i32),
                ("scale", c_f32), ("causal", c_i32), ("v16", c_i32), ("nsplit", c_i32), ("scratch", c_vp),
                ("rope_cos", c_vp), ("rope_sin", c_vp), ("k_new", c_vp), ("v_new", c_vp)]


class OneshotArgs(C.Structure):
    _fields_ = [("data", c_vp), ("gather_out", c_vp), ("stage", c_vp), ("flags", c_vp), ("epoch", c_vp), ("status", c_vp),
                ("n", c_i32), ("cap", c_i32), ("rank", c_i32), ("world", c_i32), ("max_spin", C.c_uint32), ("chunk", c_i32),
                ("mode", c_i32), ("reserved", c_i32)]


class AttnDecodeArgs(C.Structure):
    _fields_ = [("qkv", c_vp), ("kcache", c_vp), ("vcache", c_vp), ("out", c_vp), ("scratch", c_vp), ("counters", c_vp),
                ("cos_tab", c_vp), ("sin_tab", c_vp), ("pos_dev", c_vp),
                ("G", c_i32), ("H", c_i32), ("D", c_i32), ("Tmax", c_i32), ("nsplit", c_i32), ("dtype", c_i32),
                ("cache_seq_stride", c_i64), ("scale", c_f32), ("reserved", c_i32)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", c_vp), ("K", c_vp), ("V", c_vp), ("O", c_vp),
                ("B", c_i32), ("H", c_i32), ("Sq", c_i32), ("Skv", c_i32), ("D", c_i32), ("reserved", c_i32),
                ("q_batch_stride", c_i64), ("q_row_stride", c_i64), ("q_head_stride", c_i64),
                ("k_batch_stride", c_i64), ("k_row_stride", c_i64), ("k_head_stride", c_i64),
                ("v_batch_stride", c_i64), ("v_row_stride", c_i64), ("v_head_stride", c_i64),
                ("o_batch_stride", c_i64), ("o_row_stride", c_i64),
                ("scale", c_f32), ("causal", c_i32), ("dtype", c_i32)]


class AttnSmallArgs(C.Structure):
    _fields_ = [("Q", c_vp), ("K", c_vp), ("V", c_vp), ("O", c_vp),
                ("B", c_i32), ("H", c_i32), ("Sq", c_i32), ("Skv", c_i32), ("D", c_i32),
                ("q_batch_stride", c_i64), ("q_row_stride", c_i64), ("q_head_stride", c_i64),
                ("k_batch_stride", c_i64), ("k_row_stride", c_i64), ("k_head_stride", c_i64),
                ("v_batch_stride", c_i64), ("v_row_stride", c_i64), ("v_head_stride", c_i64),
                ("o_batch_stride", c_i64), ("o_row_stride", c_i64),
                ("scale", c_f32), ("dtype", c_i32)]


# name -> argtypes (restype is always int unless noted). Mirrors include/seedx_hip.h one to one.
SIGNATURES = {
    "sx_version": [],
    "sx_gemm": [C.POINTER(GemmArgs), c_vp],
    "sx_gemm_gn": [C.POINTER(GemmArgs), c_vp, c_i32, c_i32, C.POINTER(c_i32), c_vp],
    "sx_gemm_ln": [C.POINTER(GemmArgs), C.POINTER(GemmLnArgs), c_vp],
    "sx_gemm_force_tile": [c_i32],
    "sx_gemm_debug_stamps": [c_vp],
    "sx_gemm_pick_tile": [c_i32, c_i32, c_i32, c_i32, c_i32],
    "sx_gemv": [C.POINTER(GemvArgs), c_vp],
    "sx_gemv_ssq_parts": [c_i32, c_i32, c_i32],
    "sx_gemv_force_valu": [c_i32],
    "sx_layernorm": [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_softmax_rows": [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_groupnorm": [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_groupnorm2": [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_groupnorm_sp": [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32,
                        c_i32, c_vp],
    "sx_attention": [C.POINTER(AttnArgs), c_vp],
    "sx_attention_variant": [c_i32],
    "sx_norm_tune": [c_i32, c_i32],
    "sx_gemv_tune": [c_i32, c_i32],
    "sx_comm_alloc": [C.POINTER(c_vp), C.c_uint64],
    "sx_comm_free": [c_vp],
    "sx_ipc_export": [c_vp, C.c_char_p],
    "sx_ipc_open": [C.c_char_p, C.POINTER(c_vp)],
    "sx_ipc_close": [c_vp],
    "sx_allreduce_oneshot": [C.POINTER(OneshotArgs), c_vp],
    "sx_halo_pack": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_attention_small": [C.POINTER(AttnSmallArgs), c_vp],
    "sx_attn_decode": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_rope_kv_append": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_rope_kv_append_b": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp],
    "sx_attn_decode_fused": [C.POINTER(AttnDecodeArgs), c_vp],
    "sx_attn_decode_b": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_f32, c_i32, c_i64, c_vp],
    "sx_greedy_next_b": [c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp],
    "sx_scatter_rows_step": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "sx_add_i32_n": [c_vp, c_i32, c_i32, c_vp],
    "sx_embedding": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "sx_scatter_rows": [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp],
    "sx_greedy_next": [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "sx_cast": [c_vp, c_i32, c_vp, c_i32, c_i64, c_vp],
    "sx_split_bf16": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp],
    "sx_split16": [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp],
    "sx_rmsnorm_planes": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_vp],
    "sx_rope_kv_append_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp],
    "sx_rope_kv_append_f32_v16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp],
    "sx_attention_f32": [C.POINTER(AttnF32Args), c_vp],
    "sx_attention_f32_variant": [c_i32],
    "sx_copy2d_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp],
    "sx_add_f32": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "sx_patchify": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_im2col3x3_small": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_avgpool_tokens": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_timestep_embedding": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "sx_nchw_to_nhwc": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp],
    "sx_nhwc_to_nchw": [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp],
    "sx_add_i32": [c_vp, c_i32, c_vp],
    "sx_profile_marker": [c_i32, c_vp],
    "sx_silu_cast": [c_vp, c_vp, c_i32, c_i64, c_vp],
    "sx_resample_u8": [c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32,
                       c_vp, c_vp],
    "sx_u8_to_chw_lut": [c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "sx_chw_to_u8_image": [c_vp, c_i32, c_i32, c_vp, c_vp],
    "sx_marker_mask": [c_vp, c_i32, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp],
    "sx_l2norm_dim1": [c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp],
    "sx_cfg_euler_step": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_f32, c_i32, c_vp],
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol. Raises if the .so or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"seedx_amd: {LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.sx_last_error.argtypes = []
    lib.sx_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().sx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")
