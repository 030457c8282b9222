"""ctypes binding of include/gipvit.h (libgipvit_hip.so).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing this module raises.  Field order / types mirror the header
one to one; ``tests/test_abi.py`` checks every declared symbol is exported.
"""
from __future__ import annotations

import ctypes as C
import os

# One HIP runtime per process: torch ships its own libamdhip64.so and the tensors whose pointers this library receives live in ITS
# context.  Loaded first, this library would pull in /opt/rocm's copy, torch a second one later, and every launch from here would fail
# with "no ROCm-capable device is detected" -- so torch goes first, whatever the importer's order.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit operand / activation format is a property of the library BUILD (include/gipvit.h gv_act_format): bfloat16 in
# libgipvit_hip.so, IEEE half in libgipvit_hip_f16.so (the same sources with -DGV_ACT_F16; --amp --amp-dtype float16).  One
# process computes in one format: GIPVIT_ACT_FORMAT picks it before the first import of this module.
ACT_FORMAT = os.environ.get("GIPVIT_ACT_FORMAT", "bf16")
if ACT_FORMAT not in ("bf16", "f16"):
    raise ImportError(f"GIPVIT_ACT_FORMAT={ACT_FORMAT!r}: 'bf16' or 'f16'")
# GIPVIT_LIB: a lab build to time against the product library (tools/lab.sh) -- lab runs never overwrite the product file
LIB_PATH = os.environ.get("GIPVIT_LIB") or os.path.join(HERE, "libgipvit_hip.so" if ACT_FORMAT == "bf16" else "libgipvit_hip_f16.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


gv_patchify_args = _struct("gv_patchify_args", [
    ("tiles", vp), ("patches", vp), ("n_img", i32), ("tile_h", i32), ("tile_w", i32), ("img_stride", i64),
    ("n_win", i32), ("win_y", i32 * 16), ("win_x", i32 * 16), ("crop", i32), ("mean", f32 * 3), ("std", f32 * 3),
    ("n_tiles", i32), ("fill", vp)])
gv_augment_params = _struct("gv_augment_params", [
    ("n_color", i32), ("order", i32 * 4), ("bf", f32), ("cf", f32), ("sf", f32), ("hue", i32), ("blur", i32), ("kc", f32), ("ks", f32),
    ("sigma", f32), ("seed", C.c_uint32), ("d4", i32), ("zoom", i32), ("a0", i32), ("a2", i32), ("cut", i32 * 4)])
gv_augment_args = _struct("gv_augment_args", [("tiles", vp), ("out", vp), ("params", vp), ("stats", vp), ("ztable", vp), ("n", i32), ("H", i32), ("W", i32)])
gv_view_params = _struct("gv_view_params", [
    ("n_color", i32), ("order", i32 * 4), ("bf", f32), ("cf", f32), ("sf", f32), ("hue", i32), ("gray", i32), ("blur", i32), ("kc", f32), ("ks", f32),
    ("solar", i32)])
gv_crop_augment_args = _struct("gv_crop_augment_args", [
    ("tiles", vp), ("out", vp), ("boxes", vp), ("params", vp), ("stats", vp), ("n_crops", i32), ("n_tiles", i32), ("tile_h", i32), ("tile_w", i32),
    ("out_size", i32)])
gv_crop_resize_args = _struct("gv_crop_resize_args", [
    ("tiles", vp), ("out", vp), ("boxes", vp), ("n_crops", i32), ("n_tiles", i32), ("tile_h", i32), ("tile_w", i32), ("out_size", i32)])
gv_layernorm_fwd_args = _struct("gv_layernorm_fwd_args", [
    ("x", vp), ("x_stride", i64), ("gamma", vp), ("beta", vp), ("y", vp), ("mean", vp), ("rstd", vp),
    ("rows", i32), ("D", i32), ("eps", f32)])
gv_layernorm_bwd_args = _struct("gv_layernorm_bwd_args", [
    ("dy", vp), ("x", vp), ("x_stride", i64), ("mean", vp), ("rstd", vp), ("gamma", vp),
    ("g", vp), ("g_stride", i64), ("gb", vp), ("gb_stride", i64), ("partials", vp),
    ("rows", i32), ("D", i32), ("g_init", i32), ("gb_scale", vp)])
gv_colsum_finalize_args = _struct("gv_colsum_finalize_args", [
    ("partials", vp), ("n_blocks", i32), ("n_which", i32), ("which", i32), ("C", i32), ("out", vp), ("accumulate", i32)])
gv_ln_finalize_args = _struct("gv_ln_finalize_args", [
    ("partials", vp), ("n_blocks", i32), ("C", i32), ("out0", vp), ("out1", vp), ("out2", vp)])
gv_colsum_args = _struct("gv_colsum_args", [
    ("x", vp), ("x_is_f32", i32), ("ld", i64), ("rows", i32), ("C", i32), ("workspace", vp), ("out", vp), ("accumulate", i32)])
gv_linear_args = _struct("gv_linear_args", [
    ("A", vp), ("B", vp), ("C", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i64), ("ldb", i64), ("ldc", i64),
    ("trans_a", i32), ("trans_b", i32), ("c_is_f32", i32), ("epilogue", i32), ("bias", vp),
    ("resid", vp), ("ldr", i64), ("aux_in", vp), ("ld_aux", i64), ("aux_out", vp), ("pos", vp), ("P", i32), ("alpha", f32), ("colsum_a", vp), ("workspace", vp), ("workspace_bytes", i64), ("row_scale", vp)])
gv_linear_ln_fwd_args = _struct("gv_linear_ln_fwd_args", [
    ("A", vp), ("W", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i64), ("ldw", i64), ("bias", vp), ("resid", vp), ("ldr", i64),
    ("out", vp), ("ldo", i64), ("gamma", vp), ("beta", vp), ("eps", f32), ("y", vp), ("mean", vp), ("rstd", vp), ("row_scale", vp)])
gv_mlp_ln_fwd_args = _struct("gv_mlp_ln_fwd_args", [
    ("A", vp), ("W1", vp), ("bias1", vp), ("W2", vp), ("bias2", vp), ("M", i32), ("N", i32), ("K", i32), ("hidden", i32),
    ("lda", i64), ("ldw1", i64), ("ldw2", i64), ("resid", vp), ("ldr", i64), ("out", vp), ("ldo", i64),
    ("gamma", vp), ("beta", vp), ("eps", f32), ("y", vp), ("mean", vp), ("rstd", vp), ("row_scale", vp)])
gv_linear_ln_bwd_args = _struct("gv_linear_ln_bwd_args", [
    ("A", vp), ("W", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i64), ("ldw", i64), ("x", vp), ("ldx", i64),
    ("mean", vp), ("rstd", vp), ("gamma", vp), ("g", vp), ("ldg", i64), ("gb", vp), ("ldgb", i64), ("partials", vp),
    ("partial_blocks", i32), ("g_init", i32), ("gb_scale", vp)])
GV_ATTN_MAX_SEG = 4
gv_attention_fwd_varlen_args = _struct("gv_attention_fwd_varlen_args", [
    ("qkv", vp), ("o", vp), ("n_seg", i32), ("n_img", i32 * GV_ATTN_MAX_SEG), ("N", i32 * GV_ATTN_MAX_SEG), ("lse", vp * GV_ATTN_MAX_SEG),
    ("H", i32), ("scale", f32), ("q_limit", i32)])
gv_attention_bwd_varlen_args = _struct("gv_attention_bwd_varlen_args", [
    ("qkv", vp), ("o", vp), ("d_o", vp), ("dqkv", vp), ("n_seg", i32), ("n_img", i32 * GV_ATTN_MAX_SEG), ("N", i32 * GV_ATTN_MAX_SEG),
    ("lse", vp * GV_ATTN_MAX_SEG), ("H", i32), ("scale", f32), ("q_limit", i32)])
gv_expand_rows_args = _struct("gv_expand_rows_args", [("per_img", vp), ("row_img", vp), ("rows", vp), ("n_rep", i32), ("n_img", i32), ("T", i32)])
GV_DW_GROUP_MAX = 4
gv_dw_problem = _struct("gv_dw_problem", [("dY", vp), ("ldy", i64), ("X", vp), ("ldx", i64), ("dW", vp), ("ldw", i64), ("colsum_dy", vp),
                                          ("M", i32), ("N", i32)])
gv_linear_dw_group_args = _struct("gv_linear_dw_group_args", [("prob", gv_dw_problem * GV_DW_GROUP_MAX), ("n", i32), ("K", i32),
                                                              ("workspace", vp), ("workspace_bytes", i64)])
gv_attention_fwd_args = _struct("gv_attention_fwd_args", [
    ("qkv", vp), ("o", vp), ("lse", vp), ("n_img", i32), ("N", i32), ("H", i32), ("scale", f32), ("q_limit", i32)])
gv_attention_bwd_args = _struct("gv_attention_bwd_args", [
    ("qkv", vp), ("o", vp), ("d_o", vp), ("lse", vp), ("dqkv", vp), ("n_img", i32), ("N", i32), ("H", i32), ("scale", f32), ("q_limit", i32)])
gv_cls_rows_args = _struct("gv_cls_rows_args", [("x", vp), ("cls", vp), ("pos", vp), ("n_img", i32), ("N", i32), ("D", i32)])
gv_tokens_bwd_args = _struct("gv_tokens_bwd_args", [
    ("g", vp), ("gpatch", vp), ("dpos", vp), ("dcls", vp), ("n_img", i32), ("N", i32), ("D", i32), ("accumulate", i32)])
gv_small_matmul_args = _struct("gv_small_matmul_args", [
    ("A", vp), ("a_is_bf16", i32), ("sam", i64), ("sak", i64), ("B", vp), ("b_is_bf16", i32), ("sbk", i64), ("sbn", i64),
    ("C", vp), ("c_is_bf16", i32), ("ldc", i64), ("bias", vp), ("M", i32), ("N", i32), ("K", i32), ("accumulate", i32)])
gv_l2norm_fwd_args = _struct("gv_l2norm_fwd_args", [("x", vp), ("y", vp), ("inv_norm", vp), ("rows", i32), ("C", i32)])
gv_l2norm_bwd_args = _struct("gv_l2norm_bwd_args", [("dy", vp), ("y", vp), ("inv_norm", vp), ("dx", vp), ("rows", i32), ("C", i32)])
gv_weightnorm_fwd_args = _struct("gv_weightnorm_fwd_args", [("v", vp), ("g", vp), ("w", vp), ("rows", i32), ("C", i32)])
gv_weightnorm_bwd_args = _struct("gv_weightnorm_bwd_args", [
    ("dw", vp), ("v", vp), ("g", vp), ("dv", vp), ("dg", vp), ("rows", i32), ("C", i32), ("accumulate", i32)])
gv_dino_loss_args = _struct("gv_dino_loss_args", [
    ("student", vp), ("teacher", vp), ("center", vp), ("dstudent", vp), ("loss", vp), ("center_sum", vp), ("workspace", vp),
    ("B", i32), ("V", i32), ("G", i32), ("K", i32), ("student_temp", f32), ("teacher_temp", f32), ("grad_scale", f32), ("hyper", vp),
    ("loss_scale", vp)])
gv_center_update_args = _struct("gv_center_update_args", [
    ("center", vp), ("center_sum", vp), ("K", i32), ("momentum", f32), ("inv_rows", f32)])
gv_softmax_lsce_args = _struct("gv_softmax_lsce_args", [
    ("logits", vp), ("target", vp), ("loss", vp), ("dlogits", vp), ("prob", vp), ("B", i32), ("C", i32), ("smoothing", f32),
    ("loss_scale", vp)])
gv_gather_cls_args = _struct("gv_gather_cls_args", [("x", vp), ("y", vp), ("n_img", i32), ("N", i32), ("D", i32)])
gv_store_f32_args = _struct("gv_store_f32_args", [("dst", vp), ("vals", f32 * 16), ("n", i32)])
gv_cast_bf16_args = _struct("gv_cast_bf16_args", [("src", vp), ("dst", vp), ("n", i64)])
gv_sumsq_args = _struct("gv_sumsq_args", [("x", vp), ("n", i64), ("workspace", vp), ("out", vp), ("accumulate", i32)])
gv_adamw_ema_args = _struct("gv_adamw_ema_args", [
    ("p", vp), ("grad", vp), ("m", vp), ("v", vp), ("p_bf16", vp), ("teacher", vp), ("teacher_bf16", vp), ("n", i64),
    ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("bias_corr1", f32), ("bias_corr2", f32),
    ("grad_scale", f32), ("clip_norm", f32), ("gnorm_sq", vp), ("teacher_momentum", f32), ("hyper", vp), ("mode", i32), ("clip_value", f32),
    ("loss_scale", vp)])
gv_loss_scale_update_args = _struct("gv_loss_scale_update_args", [
    ("state", vp), ("gnorm_sq", vp), ("growth_factor", f32), ("backoff_factor", f32), ("growth_interval", i32)])
u32 = C.c_uint32
gv_dropout_args = _struct("gv_dropout_args", [("x", vp), ("x_is_f32", i32), ("n", i64), ("seed", u32), ("threshold", u32), ("scale", f32)])
gv_dropout_add_args = _struct("gv_dropout_add_args", [("t", vp), ("resid", vp), ("out", vp), ("row_scale", vp), ("rows", i32), ("cols", i32),
                                                      ("seed", u32), ("threshold", u32), ("scale", f32)])
gv_agc_args = _struct("gv_agc_args", [("p", vp), ("grad", vp), ("units", vp), ("n_units", i32), ("clip_factor", f32), ("eps", f32), ("grad_scale", f32)])
gv_lamb_args = _struct("gv_lamb_args", [
    ("p", vp), ("grad", vp), ("m", vp), ("v", vp), ("p_bf16", vp), ("teacher", vp), ("teacher_bf16", vp), ("blocks", vp), ("n_blocks", i32),
    ("stats", vp), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("bias_corr1", f32), ("bias_corr2", f32),
    ("grad_scale", f32), ("clip_norm", f32), ("max_grad_norm", f32), ("gnorm_sq", vp), ("teacher_momentum", f32), ("phase", i32)])

# entry point -> argument struct (every `int gv_*(const args*, void* stream)` of the header)
ENTRY_POINTS = {
    "gv_patchify": gv_patchify_args, "gv_crop_resize": gv_crop_resize_args, "gv_crop_augment": gv_crop_augment_args, "gv_augment": gv_augment_args, "gv_layernorm_fwd": gv_layernorm_fwd_args, "gv_layernorm_bwd": gv_layernorm_bwd_args,
    "gv_colsum_finalize": gv_colsum_finalize_args, "gv_ln_finalize": gv_ln_finalize_args, "gv_colsum": gv_colsum_args, "gv_linear": gv_linear_args,
    "gv_linear_ln_fwd": gv_linear_ln_fwd_args, "gv_mlp_ln_fwd": gv_mlp_ln_fwd_args, "gv_linear_ln_bwd": gv_linear_ln_bwd_args, "gv_expand_rows": gv_expand_rows_args, "gv_linear_dw_group": gv_linear_dw_group_args,
    "gv_attention_fwd": gv_attention_fwd_args, "gv_attention_fwd_varlen": gv_attention_fwd_varlen_args, "gv_attention_bwd": gv_attention_bwd_args, "gv_attention_bwd_varlen": gv_attention_bwd_varlen_args, "gv_cls_rows": gv_cls_rows_args,
    "gv_tokens_bwd": gv_tokens_bwd_args, "gv_small_matmul": gv_small_matmul_args, "gv_l2norm_fwd": gv_l2norm_fwd_args,
    "gv_l2norm_bwd": gv_l2norm_bwd_args, "gv_weightnorm_fwd": gv_weightnorm_fwd_args, "gv_weightnorm_bwd": gv_weightnorm_bwd_args,
    "gv_dino_loss": gv_dino_loss_args, "gv_center_update": gv_center_update_args, "gv_softmax_lsce": gv_softmax_lsce_args,
    "gv_gather_cls": gv_gather_cls_args, "gv_cast_bf16": gv_cast_bf16_args, "gv_store_f32": gv_store_f32_args, "gv_sumsq": gv_sumsq_args,
    "gv_adamw_ema": gv_adamw_ema_args, "gv_loss_scale_update": gv_loss_scale_update_args, "gv_lamb": gv_lamb_args, "gv_agc": gv_agc_args, "gv_dropout": gv_dropout_args, "gv_dropout_add": gv_dropout_add_args,
    # fp32 operand mode: the same structs with every bf16 buffer read / written as f32
    "gv_linear_f32": gv_linear_args, "gv_attention_fwd_f32": gv_attention_fwd_args, "gv_attention_bwd_f32": gv_attention_bwd_args,
    "gv_layernorm_fwd_f32": gv_layernorm_fwd_args, "gv_layernorm_bwd_f32": gv_layernorm_bwd_args, "gv_patchify_f32": gv_patchify_args,
    "gv_tokens_bwd_f32": gv_tokens_bwd_args, "gv_l2norm_fwd_f32": gv_l2norm_fwd_args, "gv_l2norm_bwd_f32": gv_l2norm_bwd_args,
    "gv_weightnorm_fwd_f32": gv_weightnorm_fwd_args, "gv_dino_loss_f32": gv_dino_loss_args,
}
PLAIN_SYMBOLS = ("gv_version", "gv_last_error", "gv_target", "gv_act_format", "gv_linear_workspace_bytes", "gv_workspace_bytes", "gv_linear_timing", "gv_linear_timing_read",
                 "gv_linear_ln_blocks")


class gv_linear_timing_row(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int32), ("seconds", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]

EPI_BIAS, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_ACCUM, EPI_POS, EPI_SAVE_PRE = 1, 2, 4, 8, 16, 32, 64
LN_PARTIAL_BLOCKS = 1024
OP_LINEAR, OP_LINEAR_DW_GROUP = 0, 1
HYP_LR, HYP_WD, HYP_BC1, HYP_BC2, HYP_TEACHER_MOM, HYP_GRAD_SCALE, HYP_TEACHER_TEMP, HYP_STUDENT_TEMP, HYP_COUNT = range(9)


class GipvitError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, st in ENTRY_POINTS.items():
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(st), vp]
        fn.restype = C.c_int
    lib.gv_version.restype = C.c_int
    lib.gv_last_error.restype = C.c_char_p
    lib.gv_target.restype = C.c_char_p
    lib.gv_act_format.restype = C.c_int
    if lib.gv_act_format() != {"bf16": 0, "f16": 1}[ACT_FORMAT]:
        raise ImportError(f"{LIB_PATH} was built for 16-bit format {lib.gv_act_format()} (0 = bf16, 1 = f16), GIPVIT_ACT_FORMAT asks for {ACT_FORMAT}")
    lib.gv_linear_workspace_bytes.restype = C.c_int64
    lib.gv_workspace_bytes.argtypes = [C.c_int32, vp]
    lib.gv_workspace_bytes.restype = C.c_int64
    lib.gv_linear_timing.argtypes = [C.c_int]
    lib.gv_linear_timing.restype = C.c_int
    lib.gv_linear_timing_read.argtypes = [C.POINTER(gv_linear_timing_row), C.c_int]
    lib.gv_linear_timing_read.restype = C.c_int
    lib.gv_linear_ln_blocks.argtypes = [C.c_int32]
    lib.gv_linear_ln_blocks.restype = C.c_int
    return lib


lib = _load()


def call(name: str, args, stream: int) -> None:
    """Invoke an entry point; raise GipvitError with gv_last_error() on failure."""
    rc = getattr(lib, name)(C.byref(args), vp(stream))
    if rc != 0:
        raise GipvitError(f"{name} failed (rc={rc}): {lib.gv_last_error().decode()}")
