"""Tensor-level wrappers over the C ABI: pass ``tensor.data_ptr()`` + the current
HIP stream.  PyTorch is plumbing here (device memory, streams); all arithmetic
runs in libgipvit_hip.so.  Every function launches asynchronously."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib as L

# the 16-bit tensor dtype of this process's library build (_lib.ACT_FORMAT): bfloat16 by default, float16 for the -DGV_ACT_F16
# library (--amp --amp-dtype float16); the name follows the default build
bf16, f32 = (torch.bfloat16 if L.ACT_FORMAT == "bf16" else torch.float16), torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _sfx(t: torch.Tensor) -> str:
    """Entry-point suffix of the fp32 operand mode: chosen by the dtype of the buffer that is bf16 on the training path."""
    if t.dtype == f32:
        return "_f32"
    if t.dtype != bf16:
        raise TypeError(f"expected a bf16 (training path) or f32 (fp32 parity mode) tensor, got {t.dtype}")
    return ""


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected a {dtype} device tensor, got {t.dtype} on {t.device}")


def augment(tiles_u8: torch.Tensor, params: torch.Tensor, stats: torch.Tensor, ztable: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Apply one augmentation record per tile on the device (gv_augment).  params: uint8 device tensor holding n packed
    gv_augment_params records (gipvit.augment packs them); stats: int64 device scratch [n]; ztable: f32 [1024]."""
    _chk(tiles_u8, torch.uint8, "tiles")
    n, H, W, _ = tiles_u8.shape
    assert params.is_cuda and params.dtype == torch.uint8 and params.numel() == n * C_sizeof_augment_params() and stats.numel() >= n
    if out is None:
        out = torch.empty_like(tiles_u8)
    a = L.gv_augment_args(tiles_u8.data_ptr(), out.data_ptr(), params.data_ptr(), stats.data_ptr(), ztable.data_ptr(), n, H, W)
    L.call("gv_augment", a, _stream())
    return out


def C_sizeof_augment_params() -> int:
    import ctypes
    return ctypes.sizeof(L.gv_augment_params)


def patchify(tiles_u8: torch.Tensor, windows: Sequence[Sequence[int]], crop: int, mean, std,
             out: Optional[torch.Tensor] = None, fill: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tiles_u8 [n_tiles, H, W, 3] u8 NHWC; windows [(y0, x0)] of side ``crop``.
    Returns bf16 patches [(len(windows) * n_tiles) * (crop/16)^2, 768], images crop-major."""
    _chk(tiles_u8, torch.uint8, "tiles")
    assert tiles_u8.dim() == 4 and tiles_u8.shape[-1] == 3 and tiles_u8.is_contiguous()
    n_tiles, H, W, _ = tiles_u8.shape
    n_img = n_tiles * len(windows)
    P = (crop // 16) ** 2
    if out is None:
        out = torch.empty(n_img * P, 768, dtype=bf16, device=tiles_u8.device)
    a = L.gv_patchify_args()
    a.tiles, a.patches, a.n_img, a.tile_h, a.tile_w = tiles_u8.data_ptr(), out.data_ptr(), n_img, H, W
    a.img_stride, a.n_win, a.crop, a.n_tiles = H * W * 3, len(windows), crop, n_tiles
    for i, (y, x) in enumerate(windows):
        a.win_y[i], a.win_x[i] = int(y), int(x)
    for c in range(3):
        a.mean[c], a.std[c] = float(mean[c]), float(std[c])
    if fill is not None:        # f32 [n_tiles, 8] device: normalised fill boxes (Cutout after Normalize, MeanPixelRegularization)
        assert fill.is_cuda and fill.dtype == f32 and fill.shape == (n_tiles, 8) and fill.is_contiguous()
        a.fill = fill.data_ptr()
    L.call("gv_patchify" + _sfx(out), a, _stream())
    return out


def crop_resize(tiles_u8: torch.Tensor, boxes: torch.Tensor, out_size: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Random-resized crops on the device: tiles_u8 [n_tiles, H, W, 3] u8, boxes int32 [n, 6] =
    (tile, y0, x0, h, w, flip) on the same device -> u8 [n, out_size, out_size, 3]; see gv_crop_resize."""
    _chk(tiles_u8, torch.uint8, "tiles")
    assert tiles_u8.dim() == 4 and tiles_u8.shape[-1] == 3 and tiles_u8.is_contiguous()
    assert boxes.dtype == torch.int32 and boxes.dim() == 2 and boxes.shape[1] == 6 and boxes.is_contiguous() and boxes.device == tiles_u8.device
    n = boxes.shape[0]
    if out is None:
        out = torch.empty(n, out_size, out_size, 3, dtype=torch.uint8, device=tiles_u8.device)
    a = L.gv_crop_resize_args(tiles_u8.data_ptr(), out.data_ptr(), boxes.data_ptr(), n, tiles_u8.shape[0], tiles_u8.shape[1],
                              tiles_u8.shape[2], out_size)
    L.call("gv_crop_resize", a, _stream())
    return out


_VIEW_PARAMS_BYTES = __import__("ctypes").sizeof(L.gv_view_params)      # one packed gv_view_params record


def crop_augment(tiles_u8: torch.Tensor, boxes: torch.Tensor, params: torch.Tensor, stats: torch.Tensor, out_size: int,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DINO views in one pass: the random-resized crop of ``crop_resize`` with each crop's ColorJitter / grayscale / 3x3 blur /
    solarise applied before the single store.  ``params``: uint8 device tensor holding n packed gv_view_params records
    (multicrop.ViewAugmentSampler.pack); ``stats``: int64 device scratch [>= n]; see gv_crop_augment."""
    _chk(tiles_u8, torch.uint8, "tiles")
    assert tiles_u8.dim() == 4 and tiles_u8.shape[-1] == 3 and tiles_u8.is_contiguous()
    assert boxes.dtype == torch.int32 and boxes.dim() == 2 and boxes.shape[1] == 6 and boxes.is_contiguous() and boxes.device == tiles_u8.device
    n = boxes.shape[0]
    assert params.dtype == torch.uint8 and params.numel() == n * _VIEW_PARAMS_BYTES and params.device == tiles_u8.device
    assert stats.dtype == torch.int64 and stats.numel() >= n and stats.device == tiles_u8.device
    if out is None:
        out = torch.empty(n, out_size, out_size, 3, dtype=torch.uint8, device=tiles_u8.device)
    a = L.gv_crop_augment_args(tiles_u8.data_ptr(), out.data_ptr(), boxes.data_ptr(), params.data_ptr(), stats.data_ptr(), n, tiles_u8.shape[0],
                               tiles_u8.shape[1], tiles_u8.shape[2], out_size)
    L.call("gv_crop_augment", a, _stream())
    return out


def layernorm_fwd(x, gamma, beta, rows: int, D: int, x_stride: Optional[int] = None, eps: float = 1e-6,
                  y=None, mean=None, rstd=None):
    _chk(x, f32, "x")
    dev = x.device
    y = torch.empty(rows, D, dtype=bf16, device=dev) if y is None else y
    mean = torch.empty(rows, dtype=f32, device=dev) if mean is None else mean
    rstd = torch.empty(rows, dtype=f32, device=dev) if rstd is None else rstd
    a = L.gv_layernorm_fwd_args(x.data_ptr(), D if x_stride is None else x_stride, gamma.data_ptr(), beta.data_ptr(),
                                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, eps)
    L.call("gv_layernorm_fwd" + _sfx(y), a, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, g, gb, partials, rows: int, D: int, x_stride=None, g_stride=None,
                  gb_stride=None, g_init: bool = False, gb_scale=None):
    """``gb_scale`` f32 [rows]: gb = bf16(g * gb_scale[row]) (stochastic depth of the branch in front of this residual add)."""
    a = L.gv_layernorm_bwd_args(dy.data_ptr(), x.data_ptr(), D if x_stride is None else x_stride, mean.data_ptr(),
                                rstd.data_ptr(), gamma.data_ptr(), g.data_ptr(), D if g_stride is None else g_stride,
                                _p(gb), D if gb_stride is None else gb_stride, partials.data_ptr(), rows, D, int(g_init), _p(gb_scale))
    if gb is not None and gb.dtype != dy.dtype:
        raise TypeError(f"layernorm_bwd: dy is {dy.dtype} but gb is {gb.dtype}")
    L.call("gv_layernorm_bwd" + _sfx(dy), a, _stream())


def colsum_finalize(partials, n_blocks: int, n_which: int, which: int, C: int, out, accumulate: bool):
    a = L.gv_colsum_finalize_args(partials.data_ptr(), n_blocks, n_which, which, C, out.data_ptr(), int(accumulate))
    L.call("gv_colsum_finalize", a, _stream())


def ln_finalize(partials, n_blocks: int, C: int, out0, out1, out2):
    """out_w += column sums of partials[:, w, :] (w = 0, 1, 2); None outputs are skipped."""
    L.call("gv_ln_finalize", L.gv_ln_finalize_args(partials.data_ptr(), n_blocks, C, _p(out0), _p(out1), _p(out2)), _stream())


def colsum(x, rows: int, C: int, workspace, out, accumulate: bool = False, ld: Optional[int] = None):
    a = L.gv_colsum_args(x.data_ptr(), int(x.dtype == f32), C if ld is None else ld, rows, C, workspace.data_ptr(),
                         out.data_ptr(), int(accumulate))
    L.call("gv_colsum", a, _stream())


def linear(A, B, C, M: int, N: int, K: int, *, trans_a=False, trans_b=False, epilogue=0, bias=None, resid=None,
           aux_in=None, aux_out=None, pos=None, P=0, alpha=1.0, lda=None, ldb=None, ldc=None, ldr=None, ld_aux=None,
           colsum_a=None, workspace=None, row_scale=None):
    """C[M,N] = op(A) op(B) (+ epilogue); see include/gipvit.h gv_linear."""
    a = L.gv_linear_args()
    a.A, a.B, a.C, a.M, a.N, a.K = A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K
    a.lda = (M if trans_a else K) if lda is None else lda
    a.ldb = (N if trans_b else K) if ldb is None else ldb
    a.ldc = N if ldc is None else ldc
    a.trans_a, a.trans_b, a.c_is_f32, a.epilogue = int(trans_a), int(trans_b), int(C.dtype == f32), epilogue
    a.bias, a.resid, a.ldr = _p(bias), _p(resid), (N if ldr is None else ldr)
    a.aux_in, a.ld_aux, a.aux_out = _p(aux_in), (N if ld_aux is None else ld_aux), _p(aux_out)
    a.pos, a.P, a.alpha, a.colsum_a, a.row_scale = _p(pos), P, alpha, _p(colsum_a), _p(row_scale)
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if A.dtype == f32:      # fp32 operand mode: A, B, C and the aux buffers are all f32
        for t in (B, C, aux_in, aux_out):
            if t is not None and t.dtype != f32:
                raise TypeError(f"linear: f32 A with a {t.dtype} operand (the fp32 mode is f32 throughout)")
    else:
        for t in (B, aux_in, aux_out):
            if t is not None and t.dtype != bf16:
                raise TypeError(f"linear: bf16 A with a {t.dtype} operand")
    L.call("gv_linear" + _sfx(A), a, _stream())
    return C


def linear_timing(enable: bool):
    """Start (True) / stop (False) HIP-event timing of every GEMM launch inside gv_linear."""
    rc = L.lib.gv_linear_timing(int(enable))
    if rc != 0:
        raise L.GipvitError(f"gv_linear_timing: {rc}")


def linear_timing_read():
    """Rows {kernel, launches, seconds, flops, bytes} per GEMM-class kernel since linear_timing(True)."""
    rows = (L.gv_linear_timing_row * 64)()
    n = L.lib.gv_linear_timing_read(rows, 64)
    if n < 0 or n > 64:
        raise L.GipvitError(f"gv_linear_timing_read: {n}: {L.lib.gv_last_error().decode()}")
    return [{"kernel": r.name.decode(), "launches": r.launches, "seconds": r.seconds, "flops": r.flops, "bytes": r.bytes} for r in rows[:n]]


def linear_ln_fwd(A, W, out, M: int, K: int, *, bias=None, resid=None, gamma=None, beta=None, y=None, mean=None, rstd=None, eps: float = 1e-6,
                  N: int = 384, row_scale=None):
    """out = A W^T + bias + resid (f32) and, with gamma, y / mean / rstd = LayerNorm of the new row; see gv_linear_ln_fwd."""
    a = L.gv_linear_ln_fwd_args(A.data_ptr(), W.data_ptr(), M, N, K, K, K, _p(bias), _p(resid), N, out.data_ptr(), N,
                                _p(gamma), _p(beta), eps, _p(y), _p(mean), _p(rstd), _p(row_scale))
    L.call("gv_linear_ln_fwd", a, _stream())


def mlp_ln_fwd(A, W1, b1, W2, out, M: int, K: int, hidden: int, *, bias2=None, resid=None, gamma=None, beta=None, y=None, mean=None, rstd=None,
               eps: float = 1e-6, N: int = 384, row_scale=None):
    """out = resid + row_scale (GELU(A W1^T + b1) W2^T + bias2) (f32) and, with gamma, y / mean / rstd = LayerNorm of the new row, in one
    launch; the hidden activation is not stored (forward-only passes); see gv_mlp_ln_fwd."""
    a = L.gv_mlp_ln_fwd_args(A.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), _p(bias2), M, N, K, hidden, K, K, hidden,
                             _p(resid), N, out.data_ptr(), N, _p(gamma), _p(beta), eps, _p(y), _p(mean), _p(rstd), _p(row_scale))
    L.call("gv_mlp_ln_fwd", a, _stream())


def linear_ln_bwd(dY, W, x, mean, rstd, gamma, g, gb, partials, M: int, K: int, *, g_init: bool = False, N: int = 384, gb_scale=None) -> int:
    """dXn = dY W (W stored [K, N]) fused with the LayerNorm backward it feeds; returns the number of partial blocks
    written (the n_blocks argument of ln_finalize); see gv_linear_ln_bwd."""
    a = L.gv_linear_ln_bwd_args(dY.data_ptr(), W.data_ptr(), M, N, K, K, N, x.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(),
                                gamma.data_ptr(), g.data_ptr(), N, _p(gb), N, partials.data_ptr(), partials.shape[0], int(g_init), _p(gb_scale))
    L.call("gv_linear_ln_bwd", a, _stream())
    return L.lib.gv_linear_ln_blocks(M)


def linear_dw_group(problems, K: int, workspace):
    """problems: [(dY [K, M] bf16, X [K, N] bf16, dW [M, N] f32 (accumulated), colsum_dy [M] f32 or None), ...] (at most 4) that
    reduce over the same K token rows -- one split-K launch + one reduce for all of them; see gv_linear_dw_group."""
    a = L.gv_linear_dw_group_args()
    a.n, a.K = len(problems), K
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    for q, (dY, X, dW, cs) in enumerate(problems):
        pr = a.prob[q]
        pr.dY, pr.ldy, pr.X, pr.ldx, pr.dW, pr.ldw = dY.data_ptr(), dY.shape[1], X.data_ptr(), X.shape[1], dW.data_ptr(), dW.shape[1]
        pr.colsum_dy, pr.M, pr.N = _p(cs), dY.shape[1], X.shape[1]
    L.call("gv_linear_dw_group", a, _stream())


def attention_fwd(qkv, n_img: int, N: int, H: int, scale: float, o=None, lse=None, q_limit: int = 0):
    dev = qkv.device
    o = torch.empty(n_img * N, H * 64, dtype=qkv.dtype, device=dev) if o is None else o
    lse = torch.empty(n_img, H, N, dtype=f32, device=dev) if lse is None else lse
    a = L.gv_attention_fwd_args(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), n_img, N, H, scale, q_limit)
    if o.dtype != qkv.dtype:
        raise TypeError(f"attention_fwd: qkv is {qkv.dtype} but o is {o.dtype}")
    L.call("gv_attention_fwd" + _sfx(qkv), a, _stream())
    return o, lse


def attention_fwd_varlen(qkv, o, segments, H: int, scale: float, q_limit: int = 0):
    """All segments of a token-concatenated row space in one call: ``segments`` = [(n_img, N, lse f32 [n_img, H, N]), ...] in row
    order; qkv [T, 3 H 64], o [T, H 64].  bf16: gv_attention_fwd_varlen (a long + a short segment share ONE launch); the fp32
    operand mode runs one call per segment."""
    if qkv.dtype != bf16 or len(segments) > L.GV_ATTN_MAX_SEG:
        row = 0
        for n_img, N, lse in segments:
            attention_fwd(qkv[row:row + n_img * N], n_img, N, H, scale, o=o[row:row + n_img * N], lse=lse, q_limit=q_limit)
            row += n_img * N
        return
    if o.dtype != qkv.dtype:
        raise TypeError(f"attention_fwd_varlen: qkv is {qkv.dtype} but o is {o.dtype}")
    a = L.gv_attention_fwd_varlen_args()
    a.qkv, a.o, a.n_seg, a.H, a.scale, a.q_limit = qkv.data_ptr(), o.data_ptr(), len(segments), H, scale, q_limit
    rows = 0
    for i, (n_img, N, lse) in enumerate(segments):
        assert lse.dtype == f32 and lse.numel() >= n_img * H * N
        a.n_img[i], a.N[i], a.lse[i] = n_img, N, lse.data_ptr()
        rows += n_img * N
    assert qkv.shape[0] >= rows and o.shape[0] >= rows and qkv.is_contiguous() and o.is_contiguous()
    L.call("gv_attention_fwd_varlen", a, _stream())


def attention_bwd(qkv, o, d_o, lse, n_img: int, N: int, H: int, scale: float, dqkv=None, q_limit: int = 0):
    """``q_limit`` > 0: d_o is zero behind the first q_limit query rows of every image -- those queries are skipped (gv_attention_bwd)."""
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    a = L.gv_attention_bwd_args(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), n_img, N, H, scale, q_limit)
    if not (o.dtype == d_o.dtype == dqkv.dtype == qkv.dtype):
        raise TypeError("attention_bwd: qkv / o / d_o / dqkv must share one dtype")
    L.call("gv_attention_bwd" + _sfx(qkv), a, _stream())
    return dqkv


def attention_bwd_varlen(qkv, o, d_o, dqkv, segments, H: int, scale: float, q_limit: int = 0):
    """Backward of all segments of a token-concatenated row space in one call: ``segments`` = [(n_img, N, lse), ...] as for
    attention_fwd_varlen; qkv / dqkv [T, 3 H 64], o / d_o [T, H 64].  bf16: gv_attention_bwd_varlen; the fp32 operand mode runs
    one call per segment."""
    if qkv.dtype != bf16 or len(segments) > L.GV_ATTN_MAX_SEG:
        row = 0
        for n_img, N, lse in segments:
            r = slice(row, row + n_img * N)
            attention_bwd(qkv[r], o[r], d_o[r], lse, n_img, N, H, scale, dqkv=dqkv[r], q_limit=q_limit)
            row += n_img * N
        return dqkv
    if not (o.dtype == d_o.dtype == dqkv.dtype == qkv.dtype):
        raise TypeError("attention_bwd_varlen: qkv / o / d_o / dqkv must share one dtype")
    a = L.gv_attention_bwd_varlen_args()
    a.qkv, a.o, a.d_o, a.dqkv, a.n_seg, a.H, a.scale = qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(), len(segments), H, scale
    a.q_limit = q_limit
    rows = 0
    for i, (n_img, N, lse) in enumerate(segments):
        assert lse.dtype == f32 and lse.numel() >= n_img * H * N
        a.n_img[i], a.N[i], a.lse[i] = n_img, N, lse.data_ptr()
        rows += n_img * N
    assert all(t.shape[0] >= rows and t.is_contiguous() for t in (qkv, o, d_o, dqkv))
    L.call("gv_attention_bwd_varlen", a, _stream())
    return dqkv


def expand_rows(per_img, row_img, rows, n_rep: int, n_img: int, T: int):
    """rows[r, t] = per_img[r, row_img[t]]: per-image stochastic-depth factors -> one factor per token row (n_rep branches)."""
    L.call("gv_expand_rows", L.gv_expand_rows_args(per_img.data_ptr(), row_img.data_ptr(), rows.data_ptr(), n_rep, n_img, T), _stream())


def cls_rows(x, cls, pos, n_img: int, N: int, D: int):
    L.call("gv_cls_rows", L.gv_cls_rows_args(x.data_ptr(), cls.data_ptr(), pos.data_ptr(), n_img, N, D), _stream())


def tokens_bwd(g, gpatch, dpos, dcls, n_img: int, N: int, D: int, accumulate: bool):
    a = L.gv_tokens_bwd_args(g.data_ptr(), gpatch.data_ptr(), dpos.data_ptr(), _p(dcls), n_img, N, D, int(accumulate))
    L.call("gv_tokens_bwd" + _sfx(gpatch), a, _stream())


def small_matmul(A, B, C, M: int, N: int, K: int, *, sam, sak, sbk, sbn, ldc=None, bias=None, accumulate=False):
    """C[m,n] (+)= sum_k A[m*sam + k*sak] B[k*sbk + n*sbn] (+ bias[n]); element strides."""
    a = L.gv_small_matmul_args(A.data_ptr(), int(A.dtype == bf16), sam, sak, B.data_ptr(), int(B.dtype == bf16), sbk, sbn,
                               C.data_ptr(), int(C.dtype == bf16), N if ldc is None else ldc, _p(bias), M, N, K, int(accumulate))
    L.call("gv_small_matmul", a, _stream())


def l2norm_fwd(x, y, inv_norm, rows: int, C: int):
    L.call("gv_l2norm_fwd" + _sfx(y), L.gv_l2norm_fwd_args(x.data_ptr(), y.data_ptr(), inv_norm.data_ptr(), rows, C), _stream())


def l2norm_bwd(dy, y, inv_norm, dx, rows: int, C: int):
    if dx.dtype != y.dtype:
        raise TypeError(f"l2norm_bwd: y is {y.dtype} but dx is {dx.dtype}")
    L.call("gv_l2norm_bwd" + _sfx(y), L.gv_l2norm_bwd_args(dy.data_ptr(), y.data_ptr(), inv_norm.data_ptr(), dx.data_ptr(), rows, C), _stream())


def weightnorm_fwd(v, g, w, rows: int, C: int):
    L.call("gv_weightnorm_fwd" + _sfx(w), L.gv_weightnorm_fwd_args(v.data_ptr(), g.data_ptr(), w.data_ptr(), rows, C), _stream())


def weightnorm_bwd(dw, v, g, dv, dg, rows: int, C: int, accumulate: bool):
    a = L.gv_weightnorm_bwd_args(dw.data_ptr(), v.data_ptr(), g.data_ptr(), dv.data_ptr(), _p(dg), rows, C, int(accumulate))
    L.call("gv_weightnorm_bwd", a, _stream())


def dino_loss(student, teacher, center, dstudent, loss, center_sum, workspace, B: int, V: int, G: int, K: int,
              student_temp: float, teacher_temp: float, grad_scale: float = 1.0, hyper=None, loss_scale=None):
    """``loss_scale``: device f32 scalar (LossScaler.state) that multiplies the gradient -- fp16 loss scaling."""
    a = L.gv_dino_loss_args(student.data_ptr(), teacher.data_ptr(), center.data_ptr(), dstudent.data_ptr(), loss.data_ptr(),
                            center_sum.data_ptr(), workspace.data_ptr(), B, V, G, K, student_temp, teacher_temp, grad_scale, _p(hyper),
                            _p(loss_scale))
    L.call("gv_dino_loss" + _sfx(dstudent), a, _stream())


def center_update(center, center_sum, K: int, momentum: float, inv_rows: float):
    L.call("gv_center_update", L.gv_center_update_args(center.data_ptr(), center_sum.data_ptr(), K, momentum, inv_rows), _stream())


def softmax_lsce(logits, target, loss, dlogits, prob, B: int, C: int, smoothing: float, loss_scale=None):
    a = L.gv_softmax_lsce_args(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), dlogits.data_ptr(), _p(prob), B, C, smoothing,
                               _p(loss_scale))
    L.call("gv_softmax_lsce", a, _stream())


def gather_cls(x, y, n_img: int, N: int, D: int):
    L.call("gv_gather_cls", L.gv_gather_cls_args(x.data_ptr(), y.data_ptr(), n_img, N, D), _stream())


def store_f32(dst, vals):
    """dst[i] = vals[i] (n <= 16) by a stream-ordered kernel whose arguments carry the values."""
    a = L.gv_store_f32_args()
    a.dst, a.n = dst.data_ptr(), len(vals)
    for i, v in enumerate(vals):
        a.vals[i] = float(v)
    L.call("gv_store_f32", a, _stream())


def cast_bf16(src, dst, n: Optional[int] = None):
    L.call("gv_cast_bf16", L.gv_cast_bf16_args(src.data_ptr(), dst.data_ptr(), src.numel() if n is None else n), _stream())


def sumsq(x, workspace, out, accumulate: bool = False, n: Optional[int] = None):
    a = L.gv_sumsq_args(x.data_ptr(), x.numel() if n is None else n, workspace.data_ptr(), out.data_ptr(), int(accumulate))
    L.call("gv_sumsq", a, _stream())


def adamw_ema(p, grad, m, v, p_bf16, teacher, teacher_bf16, n: int, *, lr, beta1, beta2, eps, weight_decay, step: int,
              grad_scale=1.0, clip_norm=0.0, gnorm_sq=None, teacher_momentum=0.0, hyper=None, mode=0, clip_value=0.0, loss_scale=None):
    """``loss_scale``: device f32 scalar S; the gradient is divided by it and a non-finite ``gnorm_sq`` skips the update (GradScaler.step)."""
    a = L.gv_adamw_ema_args(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p_bf16), _p(teacher), _p(teacher_bf16), n,
                            lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step,
                            grad_scale, clip_norm, _p(gnorm_sq), teacher_momentum, _p(hyper), mode, clip_value, _p(loss_scale))
    L.call("gv_adamw_ema", a, _stream())


class LossScaler:
    """torch.cuda.amp.GradScaler as timm's NativeScaler drives it (reference train.py:585-602, 1061-1070), kept on the device:
    ``state`` = f32 [S, consecutive finite steps, skipped steps, applied steps].  The loss kernels multiply their gradient by S, gv_adamw_ema divides
    it out and skips the update when the gradient's sum of squares is not finite, ``update`` is GradScaler.update().  Nothing comes
    back to the host; ``state_dict`` (checkpoints: timm's 'amp_scaler' entry) synchronises."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0], dtype=f32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    @property
    def scale(self):        # what the kernels take: the loss kernels read [0], gv_adamw_ema [0] and [3]
        return self.state

    def update(self, gnorm_sq):
        a = L.gv_loss_scale_update_args(self.state.data_ptr(), gnorm_sq.data_ptr(), self.growth_factor, self.backoff_factor, self.growth_interval)
        L.call("gv_loss_scale_update", a, _stream())

    def state_dict(self):
        s = self.state.tolist()
        return {"scale": s[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(s[1]), "skipped_steps": int(s[2]), "applied_steps": int(s[3])}

    def load_state_dict(self, d):
        self.growth_factor, self.backoff_factor = float(d["growth_factor"]), float(d["backoff_factor"])
        self.growth_interval = int(d["growth_interval"])
        self.state.copy_(torch.tensor([float(d["scale"]), float(d["_growth_tracker"]), float(d.get("skipped_steps", 0)),
                                       float(d.get("applied_steps", 0))], dtype=f32))


def lamb_block_table(tensors, chunk: int = 1 << 16) -> torch.Tensor:
    """Block table of gv_lamb for tensors = [(lo, hi), ...] (element ranges of a flat buffer, multiples of 4, in tensor-id order):
    int32 [n_blocks, 3] = (tensor, lo, hi), slices of at most ``chunk`` elements that never cross a tensor."""
    rows = []
    for t, (lo, hi) in enumerate(tensors):
        assert lo % 4 == 0 and hi % 4 == 0 and hi > lo
        for a in range(lo, hi, chunk):
            rows.append((t, a, min(hi, a + chunk)))
    return torch.tensor(rows, dtype=torch.int32)


def lamb(p, grad, m, v, p_bf16, teacher, teacher_bf16, blocks, stats, gnorm_sq, *, phase: int, lr, beta1, beta2, eps, weight_decay, step: int,
         grad_scale=1.0, clip_norm=0.0, max_grad_norm=1.0, teacher_momentum=0.0):
    """One phase of LAMB over the tensors of ``blocks`` (device int32 [n, 3], offsets relative to the buffers passed); see gv_lamb."""
    assert blocks.dtype == torch.int32 and blocks.is_contiguous() and stats.dtype == f32
    a = L.gv_lamb_args(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p_bf16), _p(teacher), _p(teacher_bf16), blocks.data_ptr(),
                       blocks.shape[0], stats.data_ptr(), lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step,
                       grad_scale, clip_norm, max_grad_norm, gnorm_sq.data_ptr(), teacher_momentum, phase)
    L.call("gv_lamb", a, _stream())


def agc_units(tensors) -> torch.Tensor:
    """Unit table of gv_agc for tensors = [(offset, shape), ...]: one unit per index of dim 0 (timm unitwise_norm: the norm over all
    other dims), the whole tensor for 1-D parameters; int32 [n_units, 2] = (offset, length)."""
    rows = []
    for off, shape in tensors:
        n = 1
        for d in shape:
            n *= d
        if len(shape) <= 1:
            rows.append((off, n))
        else:
            per = n // shape[0]
            rows.extend((off + r * per, per) for r in range(shape[0]))
    return torch.tensor(rows, dtype=torch.int32)


def agc(p, grad, units, clip_factor: float, eps: float = 1e-3, grad_scale: float = 1.0):
    """Adaptive gradient clipping in place on ``grad`` (units: device int32 [n, 2] from agc_units); see gv_agc."""
    assert units.dtype == torch.int32 and units.is_contiguous() and units.device == grad.device
    L.call("gv_agc", L.gv_agc_args(p.data_ptr(), grad.data_ptr(), units.data_ptr(), units.shape[0], clip_factor, eps, grad_scale), _stream())


def dropout_threshold(p: float) -> int:
    """keep iff hash >= floor(p * 2^32) (gv_dropout)."""
    return min(int(p * 4294967296.0), 4294967295)


def dropout_site_seed(seed: int, layer: int, site: int) -> int:
    """The seed of one dropout site of one step: murmur3 finaliser of (step seed, layer, site); sites: 0 pos-embed, 1 attn.proj,
    2 MLP activation, 3 mlp.fc2.  (Restated in oracle/vit_oracle.py for the checker.)"""
    h = (seed ^ (0x85EBCA6B * (layer * 8 + site + 1))) & 0xFFFFFFFF
    h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
    return h


def dropout(x, seed: int, p: float, n: Optional[int] = None):
    """In place on a contiguous bf16 / f32 tensor: nn.Dropout(p) with the counter-based mask of (seed, element index); see gv_dropout."""
    assert x.is_contiguous() and x.dtype in (bf16, f32)
    a = L.gv_dropout_args(x.data_ptr(), int(x.dtype == f32), x.numel() if n is None else n, seed, dropout_threshold(p), 1.0 / (1.0 - p))
    L.call("gv_dropout", a, _stream())


def dropout_add(t, resid, out, rows: int, cols: int, seed: int, p: float, row_scale=None):
    """out = resid + row_scale * dropout(t) (all f32 [rows, cols]); see gv_dropout_add."""
    assert t.dtype == f32 and resid.dtype == f32 and out.dtype == f32
    a = L.gv_dropout_add_args(t.data_ptr(), resid.data_ptr(), out.data_ptr(), _p(row_scale), rows, cols, seed, dropout_threshold(p), 1.0 / (1.0 - p))
    L.call("gv_dropout_add", a, _stream())
