/*
 * gipvit.h -- C ABI of libgipvit_hip.so: the MI355X (gfx950) implementation of the
 * ViT-encoder + DINO multi-crop training hot path of
 * noam-mosh/GipMed-Project-Self-Supervised-ViT.
 *
 * The reference has no FFI of its own for this path: the arithmetic is reached
 * through ATen library calls issued by timm's VisionTransformer
 * (reference train.py:482-495 create_model, train.py:1044-1078 step) and by the
 * orphaned nn_encoder_arch ViT / DINOHead bytecode (cited
 * below as vit.pyc@L<n>).  Each entry point names the reference call it replaces.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr());
 *     the library allocates nothing, keeps no reference after return, and only
 *     enqueues kernels on `stream` (a hipStream_t passed as void*); it never
 *     synchronises, so every call is hipGraph-capturable;
 *   - return 0 on success, <0 = GV_E_* argument error (message in
 *     gv_last_error(), thread local), >0 = hipError_t of a failed launch;
 *   - "bf16" buffers hold the library build's 16-bit format (uint16_t patterns): bfloat16 in libgipvit_hip.so, IEEE
 *     half in libgipvit_hip_f16.so -- the same sources built with -DGV_ACT_F16 for the reference's
 *     --amp --amp-dtype float16 (train.py:452-465); gv_act_format() says which.  "f32" is IEEE float;
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 */
#ifndef GIPVIT_H
#define GIPVIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GV_ABI_VERSION 9
enum { GV_HYP_LR = 0, GV_HYP_WD, GV_HYP_BC1, GV_HYP_BC2, GV_HYP_TEACHER_MOM, GV_HYP_GRAD_SCALE,
       GV_HYP_TEACHER_TEMP, GV_HYP_STUDENT_TEMP, GV_HYP_COUNT };

enum {
    GV_OK = 0,
    GV_E_SHAPE = -1,      /* unsupported / inconsistent shape            */
    GV_E_ALIGN = -2,      /* pointer or leading dimension misaligned     */
    GV_E_NULL = -3,       /* required pointer is NULL                    */
    GV_E_UNSUPPORTED = -4 /* flag / dtype combination not built          */
};

/* ---- library ---------------------------------------------------------- */
int gv_version(void);
const char* gv_last_error(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char* gv_target(void);
/* the 16-bit operand / activation format this library was built for */
enum { GV_ACT_FORMAT_BF16 = 0, GV_ACT_FORMAT_F16 = 1 };
int gv_act_format(void);

/* ---- patchify: replaces PatchEmbed's Conv2d im2col on the reference input
 * contract (vit.pyc@L167-170; datasets.py:614-631 -> transformations.py:124-128
 * ToTensor + Normalize).  Reads crop windows of NHWC uint8 tiles, applies
 * (u8/255 - mean[c]) / std[c] and writes bf16 patch rows [n_img*P, 768] with
 * k = c*256 + py*16 + px (the flatten order of Conv2d weight [D,3,16,16]). */
typedef struct {
    const uint8_t* tiles; /* [n_img, tile_h, tile_w, 3] u8 (NHWC)             */
    void* patches;        /* bf16 [n_img * (crop/16)^2, 768]                  */
    int32_t n_img;        /* images in this crop group                        */
    int32_t tile_h, tile_w;
    int64_t img_stride;   /* bytes between consecutive images                 */
    int32_t n_win;        /* crop windows per tile (images = tiles x windows) */
    int32_t win_y[16], win_x[16]; /* window origin per window index           */
    int32_t crop;         /* window side, multiple of 16                      */
    float mean[3], std[3];
    /* image index i -> tile i % n_tiles, window i / n_tiles (crop-major)     */
    int32_t n_tiles;
    /* optional device f32 [n_tiles][8] = {y0, y1, x0, x1, v_r, v_g, v_b, on}: pixels of tile t inside the box take the
     * NORMALISED value v_c -- Cutout applied after Normalize (transformations.py:206-207: zeros) and
     * MyMeanPixelRegularization (91-100: the whole tile becomes one colour) are not uint8 pixel values          */
    const float* fill;
} gv_patchify_args;
int gv_patchify(const gv_patchify_args* a, void* stream);

/* ---- random-resized crops of the tiles (DINO multi-crop input stage; absent from the
 * reference, whose tiles are augmented on the CPU by transformations.py:103-209 -- SURVEY 8f rank 1).
 * For crop n: box (y0, x0, h, w) of tile `tile` is resampled to out_size x out_size with
 * torchvision's tensor-mode semantics (float32 bilinear, align_corners=False, no antialias,
 * round half to even, clamp) and mirrored left-right when flip != 0.
 * boxes: device int32 [n_crops][6] = {tile, y0, x0, h, w, flip}; the box must lie inside the
 * tile (the caller checks: the values are device-resident).  out: u8 [n_crops, out, out, 3].   */
typedef struct {
    const uint8_t* tiles;  /* [n_tiles, tile_h, tile_w, 3] u8 (NHWC) */
    uint8_t* out;
    const int32_t* boxes;
    int32_t n_crops, n_tiles, tile_h, tile_w, out_size;
} gv_crop_resize_args;
int gv_crop_resize(const gv_crop_resize_args* a, void* stream);

/* ---- DINO view augmentation fused behind the random-resized crop: ONE pass over the tiles writes every augmented view
 * (DINO's DataAugmentationDINO -- absent from the reference; its pixel operations are the ones the reference applies to
 * whole tiles on the CPU, transformations.py:140-147: torchvision ColorJitter on PIL images + GaussianBlur(3)).  Per crop,
 * with that crop's own draws (made on the host):
 *   random-resized crop + flip (exactly gv_crop_resize's arithmetic)  ->  ColorJitter ops in `order` (n_color = 0: the
 *   RandomApply missed)  ->  RandomGrayscale (PIL "L", replicated)  ->  Gaussian blur 3x3 on the jittered view (float32,
 *   reflect padding, round half to even)  ->  solarise (PIL ImageOps.solarize: v >= threshold -> 255 - v).
 * ColorJitter's contrast blends with the mean grey of the VIEW as it is when the operation runs: a first kernel takes it
 * per crop (`stats`, device uint64 [n_crops], scratch).  The resized crop itself never exists in HBM.                      */
typedef struct {
    int32_t n_color;             /* colour operations applied (0..4)                                            */
    int32_t order[4];            /* 0 brightness, 1 contrast, 2 saturation, 3 hue, in application order          */
    float bf, cf, sf;            /* ImageEnhance factors                                                        */
    int32_t hue;                 /* added to the H byte (mod 256)                                               */
    int32_t gray;                /* 1: grayscale view                                                           */
    int32_t blur; float kc, ks;  /* 3x3 Gaussian: centre / side weight of the normalised 1-D kernel             */
    int32_t solar;               /* solarise threshold (DINO: 128), < 0: off                                    */
} gv_view_params;
typedef struct {
    const uint8_t* tiles;        /* [n_tiles, tile_h, tile_w, 3] u8 (NHWC)                                      */
    uint8_t* out;                /* [n_crops, out_size, out_size, 3] u8                                         */
    const int32_t* boxes;        /* device int32 [n_crops][6] = {tile, y0, x0, h, w, flip} (as gv_crop_resize)  */
    const gv_view_params* params; /* device [n_crops]                                                           */
    uint64_t* stats;
    int32_t n_crops, n_tiles, tile_h, tile_w, out_size;
} gv_crop_augment_args;
int gv_crop_augment(const gv_crop_augment_args* a, void* stream);

/* ---- tile augmentation on the device (replaces the CPU / PIL recipes of transformations.py:131-197 on tiles that are
 * already in HBM).  One parameter record per tile, drawn on the host; operations in reference order:
 *   colour ops in `order` (0 brightness, 1 contrast, 2 saturation, 3 hue: torchvision ColorJitter on PIL images, PIL's
 *   float32 blend with truncation; hue = +`hue` on the H byte of PIL's HSV image) -> Gaussian blur 3x3 (float32, reflect
 *   padding, round half to even; kc / ks = centre / side weight of the 1-D kernel) -> Gaussian noise x/255 + sigma * z,
 *   clip, (255 x) truncated (transformations.py:71-88; z = ztable[murmur3(seed, pixel, channel) >> 22]) -> dihedral element
 *   d4 (bit 0 transpose, bit 1 mirror rows, bit 2 mirror columns: any sequence of flips / MyRotation, 48-56) -> NEAREST
 *   zoom about the centre in 16.16 fixed point (RandomAffine(degrees=0, scale), PIL semantics) -> black box `cut`
 *   (Cutout on the [0, 1] tensor, 10-45).  Byte-exact against oracle/augment_oracle.py.
 * stats: device scratch u64 [n] (zeroed by the call); ztable: device f32 [1024], quantiles of N(0, 1).                  */
typedef struct {
    int32_t n_color; int32_t order[4];
    float bf, cf, sf; int32_t hue;
    int32_t blur; float kc, ks;
    float sigma; uint32_t seed;
    int32_t d4;
    int32_t zoom, a0, a2;
    int32_t cut[4];              /* y0, y1, x0, x1 in output coordinates; empty when y0 >= y1            */
} gv_augment_params;
typedef struct {
    const uint8_t* tiles; uint8_t* out;       /* u8 [n, H, W, 3] NHWC, out != tiles                     */
    const gv_augment_params* params;          /* device [n]                                             */
    uint64_t* stats; const float* ztable;
    int32_t n, H, W;
} gv_augment_args;
int gv_augment(const gv_augment_args* a, void* stream);

/* ---- LayerNorm (nn.LayerNorm(D, eps=1e-6); vit.pyc@L138,142,195) -------
 * fwd: x f32 rows -> y bf16 rows (+ mean, rstd f32 per row).
 * D must be 192, 384 or 768.  Row r of x lives at x + r*x_stride.        */
typedef struct {
    const float* x; int64_t x_stride;
    const float* gamma; const float* beta;
    void* y;              /* bf16 [rows, D] compact                         */
    float* mean; float* rstd; /* [rows]                                      */
    int32_t rows, D; float eps;
} gv_layernorm_fwd_args;
int gv_layernorm_fwd(const gv_layernorm_fwd_args* a, void* stream);

/* bwd: g[r] (f32, the residual-stream gradient, stride g_stride) += dLN/dx;
 * gb[r] = bf16(g[r]); partial column sums of dy*xhat, dy and new g are left
 * in `partials` [GV_LN_PARTIAL_BLOCKS, 3, D] f32 for gv_colsum_finalize.   */
#define GV_LN_PARTIAL_BLOCKS 1024
typedef struct {
    const void* dy;       /* bf16 [rows, D] compact                         */
    const float* x; int64_t x_stride;
    const float* mean; const float* rstd; const float* gamma;
    float* g; int64_t g_stride;   /* in/out                                  */
    void* gb; int64_t gb_stride;  /* bf16 out (may be NULL)                  */
    float* partials;
    int32_t rows, D;
    int32_t g_init;       /* 1: g is treated as 0 on input (g = dx)          */
    const float* gb_scale; /* optional [rows]: gb[r] = bf16(g[r] * gb_scale[r]) -- the gradient entering a branch whose
                            * output was scaled by stochastic depth; NULL = 1  */
} gv_layernorm_bwd_args;
int gv_layernorm_bwd(const gv_layernorm_bwd_args* a, void* stream);

/* out[c] (+)= sum_b partials[b, which, c] for b < n_blocks                 */
typedef struct {
    const float* partials; int32_t n_blocks, n_which, which, C;
    float* out; int32_t accumulate;
} gv_colsum_finalize_args;
int gv_colsum_finalize(const gv_colsum_finalize_args* a, void* stream);

/* all three column sums of a LayerNorm-backward partial buffer in one launch:
 * out_w[c] += sum_b partials[b, w, c] for w = 0..2 (NULL outputs are skipped).  Always
 * accumulates (row-split blocks meet in the outputs through f32 atomics).             */
typedef struct {
    const float* partials; int32_t n_blocks, C;
    float* out0; float* out1; float* out2;
} gv_ln_finalize_args;
int gv_ln_finalize(const gv_ln_finalize_args* a, void* stream);

/* column sums of a bf16 or f32 matrix [rows, C] -> out[C] f32 (bias grads,
 * train.py:1071 backward of nn.Linear bias).  workspace >= 64*C floats.    */
typedef struct {
    const void* x; int32_t x_is_f32; int64_t ld;
    int32_t rows, C;
    float* workspace; float* out; int32_t accumulate;
} gv_colsum_args;
int gv_colsum(const gv_colsum_args* a, void* stream);

/* ---- linear / GEMM (nn.Linear fwd + both backward products; vit.pyc@L98-104,
 * L119-131, L326-330).  C[M,N] = op(A) . op(B) with bf16 operands, f32 MFMA
 * accumulation (v_mfma_f32_16x16x32_bf16):
 *    trans_a = 0: A stored [M,K];  1: A stored [K,M]
 *    trans_b = 0: B stored [N,K] (torch Linear weight);  1: B stored [K,N]
 * forward  y = x W^T      : (0,0)        dX = dY W : (0,1)     dW = dY^T X : (1,1)
 * Epilogue, applied in this order to the f32 accumulator v at (m,n):
 *    BIAS: v += bias[n];  SAVE_PRE: aux_out[m,n] = bf16(v);  GELU: v = gelu(v);
 *    DGELU: v *= gelu'(aux_in[m,n]);  RESID: v += resid[m,n] (f32);
 *    POS: token-row remap m -> m + m/P + 1 and v += pos[(m%P)+1, n] (patch embed);
 *    ACCUM: v += C[m,n] (f32 out only);  then C[m,n] = v (bf16 or f32).       */
enum {
    GV_EPI_BIAS = 1, GV_EPI_GELU = 2, GV_EPI_RESID = 4, GV_EPI_DGELU = 8,
    GV_EPI_ACCUM = 16, GV_EPI_POS = 32, GV_EPI_SAVE_PRE = 64,
    GV_EPI_ALL = 127      /* any other bit is rejected with GV_E_UNSUPPORTED */
};
typedef struct {
    const void* A; const void* B; void* C;
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t trans_a, trans_b;
    int32_t c_is_f32;
    int32_t epilogue;           /* GV_EPI_* bitmask                          */
    const float* bias;          /* [N] f32                                   */
    const float* resid; int64_t ldr;   /* f32 [M,N]                          */
    const void* aux_in; int64_t ld_aux; /* bf16 [M,N] pre-activation (DGELU) */
    void* aux_out;              /* bf16 [M,N], ld = ld_aux (SAVE_PRE)        */
    const float* pos; int32_t P; /* POS: pos f32 [(P+1), N]                  */
    float alpha;                /* scales the accumulator before the epilogue*/
    /* trans_a only: colsum_a[m] += sum_k A[k,m] (f32, atomics) -- the bias gradient of the
     * Linear whose weight gradient this launch computes, from the tiles it stages anyway */
    float* colsum_a;
    /* optional split-K scratch (f32, caller-owned, gv_linear_workspace_bytes() big): when given,
     * split-K partial tiles are stored as slabs and summed into C by a second kernel instead of
     * meeting in C through f32 atomics (the atomic volume is ~33 MB per launch at full occupancy
     * and runs at ~1.3 TB/s; slab stores + reduce move the same bytes at HBM speed).           */
    float* workspace; int64_t workspace_bytes;
    /* RESID only, optional: v *= row_scale[m] before the residual add -- stochastic depth (timm DropPath, train.py:283-288
     * --drop-path: keep / (1 - p) of the token's image, see gv_expand_rows); NULL = 1                                  */
    const float* row_scale;
} gv_linear_args;
/* upper bound of the split-K scratch gv_linear can use for any shape */
int64_t gv_linear_workspace_bytes(void);
/* Scratch ONE call would use if the full gv_linear_workspace_bytes() were offered (`gv_workspace_bytes(op, shape)`):
 * op = GV_OP_LINEAR with a gv_linear_args, GV_OP_LINEAR_DW_GROUP with a gv_linear_dw_group_args -- shapes, flags, leading
 * dimensions and operand ALIGNMENT as in the real call (pointers are checked for alignment, never dereferenced; the
 * workspace fields are ignored).  The entry point's own kernel selection runs in a plan mode, nothing is launched: 0 = the call
 * takes no scratch (e.g. the full-row kernels), -1 = the arguments would be rejected (gv_last_error()).  Every other entry
 * point takes caller-sized buffers named in its struct and needs no query.                                                  */
#define GV_OP_LINEAR 0
#define GV_OP_LINEAR_DW_GROUP 1
int64_t gv_workspace_bytes(int32_t op, const void* args);
int gv_linear(const gv_linear_args* a, void* stream);

/* Weight gradients of several Linears that reduce over the SAME token rows (one transformer block: attn.qkv, attn.proj,
 * mlp.fc1, mlp.fc2; autograd of vit.pyc@L98-104, L119-131) in ONE launch:  dW_q[M_q, N_q] += dY_q^T X_q over K rows, and
 * colsum_dy_q[m] += sum_k dY_q[k, m] (the bias gradient) where given.  Split-K fills the chip ONCE for the whole group
 * instead of once per product: 108 tiles x 4 slices for a ViT-S block instead of 4 x (9..36 tiles x 14..56 slices), so the
 * partial-tile slab traffic drops from 4 x 32 MB to 28 MB and one reduce launch serves all products.
 * workspace: f32 scratch, gv_linear_workspace_bytes() big.                                                             */
#define GV_DW_GROUP_MAX 4
typedef struct {
    const void* dY; int64_t ldy;     /* bf16 [K, M]  (rows = tokens)                         */
    const void* X;  int64_t ldx;     /* bf16 [K, N]                                          */
    float* dW; int64_t ldw;          /* f32 [M, N], accumulated into                         */
    float* colsum_dy;                /* f32 [M] or NULL, accumulated into                    */
    int32_t M, N;
} gv_dw_problem;
typedef struct {
    gv_dw_problem prob[GV_DW_GROUP_MAX];
    int32_t n, K;
    float* workspace; int64_t workspace_bytes;
} gv_linear_dw_group_args;
int gv_linear_dw_group(const gv_linear_dw_group_args* a, void* stream);

/* Live per-kernel timing of the GEMM-class launches (gv_linear, gv_linear_ln_fwd, gv_linear_ln_bwd; bench.py's
 * roofline leg): while enabled, every such launch is bracketed by two HIP events on the launch stream.
 * gv_linear_timing(1) clears earlier records and starts recording, gv_linear_timing(0) stops.  _read synchronises
 * on the recorded events and folds them into one row per kernel (name = the kernel instantiation as rocprofv3
 * prints it, e.g. "gemm_kernel<true, true, float, true, 16>"); returns the number of rows written (<= max_rows)
 * or an error.                                                                                                  */
typedef struct {
    char name[96];
    int32_t launches;
    double seconds, flops;   /* summed over the launches */
    double bytes;            /* algorithmic HBM bytes summed over the launches: every operand read once, every output written once */
} gv_linear_timing_row;
int gv_linear_timing(int enable);
int gv_linear_timing_read(gv_linear_timing_row* rows, int max_rows);

/* ---- full-row Linear fused with the LayerNorm around it (ViT-S width: N must be 384) --------------------------
 * One workgroup owns whole output rows, so the row-wise LayerNorm work happens in the GEMM epilogue and the f32
 * residual row makes one HBM round trip instead of two (csrc/panel.hip).
 *
 * fwd -- replaces `x = x + Linear(a)` of Block.forward (attn.proj / mlp.fc2, vit.pyc@L146-152) TOGETHER WITH the
 * LayerNorm that reads the new x next (norm2 of the block / norm1 of the next block, vit.pyc@L138,142):
 *     out[m,:] = A[m,:] . W^T + bias + resid[m,:]          (f32; W is the Linear weight [N, K], bf16)
 *     y[m,:]   = bf16( (out[m,:] - mean[m]) * rstd[m] * gamma + beta ),  mean / rstd of out[m,:] (eps inside the sqrt)
 * gamma == NULL skips the LayerNorm outputs (last block: the final norm reads CLS rows only); bias / resid optional. */
typedef struct {
    const void* A; const void* W;        /* bf16 [M, K] (lda), bf16 [N, K] (ldw)            */
    int32_t M, N, K; int64_t lda, ldw;
    const float* bias;                   /* [N] or NULL                                     */
    const float* resid; int64_t ldr;     /* f32 [M, N] or NULL                              */
    float* out; int64_t ldo;             /* f32 [M, N]                                      */
    const float* gamma; const float* beta; float eps;
    void* y;                             /* bf16 [M, N] compact                             */
    float* mean; float* rstd;            /* [M]                                             */
    const float* row_scale;              /* optional [M]: out = resid + row_scale[m] * (A W^T + bias) (stochastic depth)    */
} gv_linear_ln_fwd_args;
int gv_linear_ln_fwd(const gv_linear_ln_fwd_args* a, void* stream);

/* Fused MLP forward (vit.pyc@L98-104 Mlp.forward inside Block.forward L146-152, for passes that save nothing: the DINO teacher,
 * inference):  out = resid + row_scale * (GELU(A W1^T + bias1) W2^T + bias2)  [f32],  and -- with gamma -- y / mean / rstd =
 * LayerNorm(out) as gv_linear_ln_fwd leaves them.  A bf16 [M, K], W1 bf16 [hidden, K], W2 bf16 [N, hidden]; N = 384,
 * K % 64 == 0, hidden % 256 == 0.  The [M, hidden] activation lives in LDS, 256 columns at a time (a K-slice of fc2);
 * results are bit-identical to gv_linear(BIAS | GELU) followed by gv_linear_ln_fwd.                                          */
typedef struct {
    const void* A; const void* W1; const float* bias1; const void* W2; const float* bias2;
    int32_t M, N, K, hidden; int64_t lda, ldw1, ldw2;
    const float* resid; int64_t ldr; float* out; int64_t ldo;
    const float* gamma; const float* beta; float eps; void* y; float* mean; float* rstd;
    const float* row_scale;
} gv_mlp_ln_fwd_args;
int gv_mlp_ln_fwd(const gv_mlp_ln_fwd_args* a, void* stream);

/* bwd -- replaces the dX product of the Linear that CONSUMED a LayerNorm output (mlp.fc1 / attn.qkv:
 * dXn = dY . W with W the Linear weight stored [K = its out features, N = 384]) together with that LayerNorm's
 * backward (autograd of vit.pyc@L138,142; gv_layernorm_bwd's contract with dy = dXn kept in f32):
 *     g[m,:] (+)= dLN/dx(dXn[m,:]; x[m,:], mean[m], rstd[m], gamma);   gb[m,:] = bf16(g[m,:])
 *     partials[b, 0..2, :] = column sums over workgroup b's rows of dXn*xhat, dXn, new g   (gv_ln_finalize folds them:
 *     dgamma, dbeta, bias gradient of the Linear in front of the residual add)
 * partials must hold gv_linear_ln_blocks(M) blocks of [3, N] f32.                                                   */
typedef struct {
    const void* A; const void* W;        /* bf16 dY [M, K] (lda), bf16 W [K, N] (ldw)       */
    int32_t M, N, K; int64_t lda, ldw;
    const float* x; int64_t ldx;         /* f32 [M, N]: the LayerNorm's input rows          */
    const float* mean; const float* rstd; const float* gamma;
    float* g; int64_t ldg;               /* f32 [M, N] in/out: residual-stream gradient     */
    void* gb; int64_t ldgb;              /* bf16 out (may be NULL)                          */
    float* partials; int32_t partial_blocks;
    int32_t g_init;                      /* 1: g is treated as 0 on input                   */
    const float* gb_scale;               /* optional [M]: gb[m,:] = bf16(g[m,:] * gb_scale[m]) (stochastic depth)           */
} gv_linear_ln_bwd_args;
int gv_linear_ln_bwd(const gv_linear_ln_bwd_args* a, void* stream);
/* workgroups (= partial blocks) a gv_linear_ln_* launch over M rows uses */
int gv_linear_ln_blocks(int32_t M);

/* ---- stochastic depth (timm DropPath behind --drop-path, train.py:283-288; vit.pyc@L66-74): n_rep sets of per-image
 * factors keep / (1 - p) (one set per residual branch) expanded to one factor per token row of a multi-crop row space:
 * rows[r*T + t] = per_img[r*n_img + row_img[t]],  row_img[t] = the image token row t belongs to.                   */
typedef struct { const float* per_img; const int32_t* row_img; float* rows; int32_t n_rep, n_img, T; } gv_expand_rows_args;
int gv_expand_rows(const gv_expand_rows_args* a, void* stream);

/* ---- attention (vit.pyc@L119-131): softmax(q k^T * scale) v per (image, head)
 * on packed qkv bf16 [n_img*N, 3, H, 64] -> o bf16 [n_img*N, H, 64];
 * lse f32 [n_img, H, N] (natural-log sum-exp of the scaled scores).
 * head_dim is fixed at 64 (ViT-T/S/B); N <= 288.                            */
typedef struct {
    const void* qkv; void* o; float* lse;
    int32_t n_img, N, H; float scale;
    int32_t q_limit;      /* > 0: only the first q_limit query rows of every image are needed (rounded up to 32): o / lse rows behind
                           * them are left untouched.  The last block of a ViT whose forward returns the CLS row (vit.pyc@L248-253)
                           * asks for 1.  0 = every query.  (The _f32 entry point computes every query.)                            */
} gv_attention_fwd_args;
int gv_attention_fwd(const gv_attention_fwd_args* a, void* stream);

/* Varlen forward (the `cu_seqlens` form): the token-concatenated row space of a multi-crop pass -- segment i
 * holds n_img[i] images of N[i] tokens, its rows start where segment i - 1 ends, qkv / o cover all segments, lse[i] is
 * segment i's f32 [n_img[i], H, N[i]].  A long (128 < N <= 224) + a short (32 < N <= 64) segment run as ONE launch
 * (the short pairs fill the long launch's last, half-empty round of workgroups); any other mix runs one launch per segment.
 * (gv_attention_bwd_varlen below takes the same segment table.)                                                              */
#define GV_ATTN_MAX_SEG 4
typedef struct {
    const void* qkv; void* o;
    int32_t n_seg; int32_t n_img[GV_ATTN_MAX_SEG]; int32_t N[GV_ATTN_MAX_SEG];
    float* lse[GV_ATTN_MAX_SEG];
    int32_t H; float scale;
    int32_t q_limit;      /* as gv_attention_fwd_args.q_limit, for every segment */
} gv_attention_fwd_varlen_args;
int gv_attention_fwd_varlen(const gv_attention_fwd_varlen_args* a, void* stream);

typedef struct {
    const void* qkv; const void* o; const void* d_o; const float* lse;
    void* dqkv;           /* bf16 [n_img*N, 3, H, 64]                        */
    int32_t n_img, N, H; float scale;
    int32_t q_limit;      /* > 0: d_o is zero behind the first q_limit query rows of every image (the caller guarantees it for the rows
                           * up to the next multiple of 32; rows behind that are not read): those queries are skipped and their dQ rows
                           * are written as zeros.  qkv / o / lse must be valid for the rows up to that multiple of 32 (what
                           * gv_attention_fwd with the same q_limit leaves).  0 = every query.                                      */
} gv_attention_bwd_args;
int gv_attention_bwd(const gv_attention_bwd_args* a, void* stream);

/* Varlen backward: the segment table of gv_attention_fwd_varlen (segment i: n_img[i] images of N[i] tokens, rows back to back in
 * qkv / o / d_o / dqkv; lse[i] as the forward left it).  One CALL for the multi-crop row space; inside, each length class runs as
 * its own launch: the long class (N > 64) holds a CU's LDS alone (114 - 142 KB per (image, head) pair) while the short classes
 * run 4 - 5 workgroups per CU, so a shared launch geometry would cost the short pairs their occupancy.  */
typedef struct {
    const void* qkv; const void* o; const void* d_o; void* dqkv;
    int32_t n_seg; int32_t n_img[GV_ATTN_MAX_SEG]; int32_t N[GV_ATTN_MAX_SEG];
    const float* lse[GV_ATTN_MAX_SEG];
    int32_t H; float scale;
    int32_t q_limit;      /* as gv_attention_bwd_args.q_limit, for every segment */
} gv_attention_bwd_varlen_args;
int gv_attention_bwd_varlen(const gv_attention_bwd_varlen_args* a, void* stream);

/* ---- token assembly (vit.pyc@L235-246 prepare_tokens: CLS row) ---------
 * x[i*N + 0, :] = cls[:] + pos[0, :] for every image i.                    */
typedef struct { float* x; const float* cls; const float* pos; int32_t n_img, N, D; } gv_cls_rows_args;
int gv_cls_rows(const gv_cls_rows_args* a, void* stream);

/* backward of token assembly: from g f32 [n_img*N, D] produce
 *   gpatch bf16 [n_img*P, D] (compact patch rows, A operand of dW_patch),
 *   dpos f32 [N, D] (+)= sum_i g[i*N + t]  (dcls = dpos row 0 before the add
 *   is also written to dcls (+)=).                                           */
typedef struct {
    const float* g; void* gpatch; float* dpos; float* dcls;
    int32_t n_img, N, D; int32_t accumulate;
} gv_tokens_bwd_args;
int gv_tokens_bwd(const gv_tokens_bwd_args* a, void* stream);

/* small strided matmul C[m,n] (+)= sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]);
 * element strides, each operand f32 or bf16.  For the tiny products of the path: the
 * bicubic pos-embed resampling (vit.pyc@L213-233) as a fixed linear map, and the
 * supervised classifier head Linear(D, num_classes) (train.py:482-495, num_classes=2). */
typedef struct {
    const void* A; int32_t a_is_bf16; int64_t sam, sak;
    const void* B; int32_t b_is_bf16; int64_t sbk, sbn;
    void* C; int32_t c_is_bf16; int64_t ldc;
    const float* bias;
    int32_t M, N, K; int32_t accumulate;
} gv_small_matmul_args;
int gv_small_matmul(const gv_small_matmul_args* a, void* stream);

/* ---- DINOHead tail (vit.pyc@L326-330) ---------------------------------
 * l2norm fwd: y bf16 = x / max(||x||, 1e-12), inv_norm f32 [rows] saved.    */
typedef struct { const float* x; void* y; float* inv_norm; int32_t rows, C; } gv_l2norm_fwd_args;
int gv_l2norm_fwd(const gv_l2norm_fwd_args* a, void* stream);
/* bwd: dx bf16 = (dy - y (y.dy)) * inv_norm  (dy f32, y bf16)               */
typedef struct { const float* dy; const void* y; const float* inv_norm; void* dx; int32_t rows, C; } gv_l2norm_bwd_args;
int gv_l2norm_bwd(const gv_l2norm_bwd_args* a, void* stream);
/* weight_norm: w bf16 [K,C] = g[k] * v[k,:] / ||v[k,:]||                    */
typedef struct { const float* v; const float* g; void* w; int32_t rows, C; } gv_weightnorm_fwd_args;
int gv_weightnorm_fwd(const gv_weightnorm_fwd_args* a, void* stream);
/* bwd: dv (+)= g/||v|| (dw - vhat (vhat.dw)); dg (+)= vhat.dw (dg may be NULL) */
typedef struct {
    const float* dw; const float* v; const float* g; float* dv; float* dg;
    int32_t rows, C; int32_t accumulate;
} gv_weightnorm_bwd_args;
int gv_weightnorm_bwd(const gv_weightnorm_bwd_args* a, void* stream);

/* ---- DINO loss (paper Alg. 1; absent from the reference)
 * student f32 [V*B, K] crop-major, teacher f32 [G*B, K] crop-major.
 * Writes loss (scalar f32, mean over the G*(V-1)... pairs and B), dstudent
 * bf16 [V*B, K] = d loss / d student * grad_scale, and center_sum f32 [K] =
 * sum over teacher rows of the raw teacher logits.
 * workspace: f32 [2 * (V+G) * B] row stats.                                 */
typedef struct {
    const float* student; const float* teacher; const float* center;
    void* dstudent; float* loss; float* center_sum; float* workspace;
    int32_t B, V, G, K;
    float student_temp, teacher_temp, grad_scale;
    const float* hyper;   /* optional device vector (see gv_adamw_ema_args): temps */
    const float* loss_scale; /* optional DEVICE scalar: dstudent is multiplied by it (fp16 loss scaling, gv_loss_scale_update) */
} gv_dino_loss_args;
int gv_dino_loss(const gv_dino_loss_args* a, void* stream);

/* center <- m * center + (1-m) * center_sum / n_rows (after the all-reduce) */
typedef struct { float* center; const float* center_sum; int32_t K; float momentum; float inv_rows; } gv_center_update_args;
int gv_center_update(const gv_center_update_args* a, void* stream);

/* ---- supervised head loss (train.py:1046 softmax, :1053 LabelSmoothingCE on
 * the soft-maxed output, gather index target[B,1] per train_instruct.txt:3-7).
 * logits f32 [B,C] (C <= 64), target i64 [B]; loss scalar; dlogits f32 [B,C].*/
typedef struct {
    const float* logits; const int64_t* target; float* loss; float* dlogits; float* prob;
    int32_t B, C; float smoothing;
    const float* loss_scale; /* optional DEVICE scalar: dlogits is multiplied by it (fp16 loss scaling); the loss itself is not */
} gv_softmax_lsce_args;
int gv_softmax_lsce(const gv_softmax_lsce_args* a, void* stream);

/* ---- gather / scatter of CLS rows: y[i,:] = x[i*N, :] (f32 -> bf16) -------*/
typedef struct { const float* x; void* y; int32_t n_img, N, D; } gv_gather_cls_args;
int gv_gather_cls(const gv_gather_cls_args* a, void* stream);

/* dst[i] = vals[i] for i < n <= 16: per-step schedule values (gv_adamw_ema_args.hyper)
 * delivered as KERNEL ARGUMENTS of a stream-ordered launch rather than by a memcpy.     */
typedef struct { float* dst; float vals[16]; int32_t n; } gv_store_f32_args;
int gv_store_f32(const gv_store_f32_args* a, void* stream);

/* f32 -> bf16 cast of a flat buffer (weight arena refresh)                  */
typedef struct { const float* src; void* dst; int64_t n; } gv_cast_bf16_args;
int gv_cast_bf16(const gv_cast_bf16_args* a, void* stream);

/* sum of squares of a flat f32 buffer -> out[0] (+)= ; workspace >= 1024 f32 */
typedef struct { const float* x; int64_t n; float* workspace; float* out; int32_t accumulate; } gv_sumsq_args;
int gv_sumsq(const gv_sumsq_args* a, void* stream);

/* ---- fused AdamW + teacher EMA + bf16 refresh over a flat parameter arena
 * (train.py:1078 optimizer.step; :1080-1081 model_ema.update).
 *   if clip_norm > 0: scale = min(1, clip_norm / (sqrt(*gnorm_sq) + 1e-6))
 *   g = grad * grad_scale * scale;  p *= 1 - lr*wd;  m,v Adam;  p -= ...
 *   p_bf16 = bf16(p);  if teacher: t = mom*t + (1-mom)*p; t_bf16 = bf16(t)   */
typedef struct {
    float* p; const float* grad; float* m; float* v; void* p_bf16;
    float* teacher; void* teacher_bf16;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2;
    float grad_scale, clip_norm; const float* gnorm_sq;
    float teacher_momentum;
    /* optional DEVICE vector; when non-NULL it overrides the by-value scalars (except
     * weight_decay, which then MULTIPLIES hyper[GV_HYP_WD]: 1 = decayed, 0 = not) so a
     * captured hipGraph can be replayed with new schedules:
     * [GV_HYP_LR, GV_HYP_WD, GV_HYP_BC1, GV_HYP_BC2, GV_HYP_TEACHER_MOM, GV_HYP_GRAD_SCALE,
     *  GV_HYP_TEACHER_TEMP, GV_HYP_STUDENT_TEMP] */
    const float* hyper;
    /* 0 = AdamW (decoupled decay, torch.optim.AdamW); 1 = Adam with L2 decay folded into the
     * gradient (timm --opt adam, the documented runs: train_instruct.txt:23); 2 = SGD with
     * Nesterov momentum beta1 and L2 decay (timm --opt sgd, the reference default, train.py:161);
     * `m` is the momentum buffer, `v` is unused in mode 2; 3 = frozen range: p, m, v are left
     * untouched (a parameter without a gradient is skipped by torch optimizers: reference
     * train.py:497-503 --no-grad, DINO's frozen last layer) and only the teacher EMA / bf16
     * refresh run.                                                                           */
    int32_t mode;
    /* > 0: every gradient element (after grad_scale) is clamped to [-clip_value, clip_value] -- `--clip-mode value`,
     * torch.nn.utils.clip_grad_value_ (reference train.py:1072-1077 dispatch_clip_grad); use instead of clip_norm */
    float clip_value;
    /* optional: the DEVICE state vector of gv_loss_scale_update, [S, .., .., applied steps] (fp16 loss scaling,
     * torch.cuda.amp.GradScaler as driven by timm's NativeScaler: reference train.py:585-602, 1061-1070).  The gradient was
     * produced from S * loss, so g = grad * grad_scale / S (the clip norm is that of the unscaled gradient); when *gnorm_sq
     * (required then) is not finite the call leaves p, m, v untouched -- GradScaler.step() skips optimizer.step() -- while the
     * teacher / --model-ema update and the 16-bit refresh still run; and Adam's bias corrections are taken at step
     * state[3] + 1 (the optimizer's own count of applied steps) instead of the by-value / hyper ones.                        */
    const float* loss_scale;
} gv_adamw_ema_args;
int gv_adamw_ema(const gv_adamw_ema_args* a, void* stream);

/* ---- GradScaler.update() (torch.cuda.amp.GradScaler: init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval
 * 2000): state[0] = S, state[1] = number of consecutive finite steps, state[2] = number of skipped steps so far (for the log),
 * state[3] = number of applied optimizer steps.
 * *gnorm_sq = sum of squares of the scaled gradient arena (after the data-parallel all-reduce, so every rank decides alike):
 *   not finite:  S *= backoff_factor, state[1] = 0, state[2] += 1
 *   finite:      state[3] += 1;  if ++state[1] == growth_interval: S *= growth_factor, state[1] = 0
 * One thread, stream-ordered behind the optimizer launches that read S; nothing returns to the host.                         */
typedef struct {
    float* state; const float* gnorm_sq;
    float growth_factor, backoff_factor; int32_t growth_interval;
} gv_loss_scale_update_args;
int gv_loss_scale_update(const gv_loss_scale_update_args* a, void* stream);

/* ---- dropout (`--drop`, reference train.py:283-284 -> create_model(drop_rate): nn.Dropout after the pos-embed add, after attn.proj,
 * after the MLP activation and after mlp.fc2; vit.pyc@L98-104, L119-131, L235-246).  Counter-based masks, so that the backward
 * regenerates what the forward used and the oracle can restate it: element i of a site is KEPT iff
 *   fmix32(seed + 0x9E3779B9 * (i + 1)) >= threshold,   threshold = floor(p * 2^32)   (murmur3 finaliser)
 * and kept values are scaled by `scale` = 1 / (1 - p).  `seed` is the site's own (host: one per step, layer and site).
 * gv_dropout: in place on x (bf16, or f32 with x_is_f32) -- an activation in the forward, the gradient entering the same site
 * in the backward.  gv_dropout_add: out[m, :] = resid[m, :] + row_scale[m] * dropout(t[m, :]) for the two residual branches
 * (t = the branch's Linear output incl. bias, f32; row_scale: stochastic depth, may be NULL).  Element index = m * cols + c.
 * These run only when --drop > 0 (the step then takes the unfused Linear / LayerNorm kernels); the default path is untouched. */
typedef struct { void* x; int32_t x_is_f32; int64_t n; uint32_t seed, threshold; float scale; } gv_dropout_args;
int gv_dropout(const gv_dropout_args* a, void* stream);
typedef struct {
    const float* t; const float* resid; float* out; const float* row_scale;
    int32_t rows, cols; uint32_t seed, threshold; float scale;
} gv_dropout_add_args;
int gv_dropout_add(const gv_dropout_add_args* a, void* stream);

/* ---- adaptive gradient clipping (`--clip-mode agc`, reference train.py:1072-1077 -> timm.utils.adaptive_clip_grad): per UNIT
 * (a row of a matrix / conv filter = dim 0 of the parameter; a whole tensor for 1-D parameters and for tensors whose dim 0 is 1)
 *   g_u <- g_u * min(1, clip_factor * max(||p_u||, eps) / max(||g_u||, 1e-6)),   g_u = grad_u * grad_scale for the norms
 * in place on the (unscaled) gradient arena, before the optimizer pass.  units: device int32 [n_units][2] = {offset, length}
 * into p / grad (offsets multiples of 4 are not required).  One wave per unit.                                            */
typedef struct {
    const float* p; float* grad; const int32_t* units; int32_t n_units;
    float clip_factor, eps, grad_scale;
} gv_agc_args;
int gv_agc(const gv_agc_args* a, void* stream);

/* ---- LAMB (timm `--opt lamb`, reference train.py:161 -> create_optimizer_v2 -> timm.optim.Lamb; named in SURVEY 8f-4) over the
 * same flat arena, two launches per range (a per-TENSOR trust ratio needs every tensor's norms before its update):
 *   phase 0:  g = grad * grad_scale * c, c = the global-norm clips (clip_norm as in gv_adamw_ema, then Lamb's own
 *             max_grad_norm: g /= max(||g|| / max_grad_norm, 1));  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
 *             u = (m / bias_corr1) / (sqrt(v) / sqrt(bias_corr2) + eps) + weight_decay * p;
 *             stats[2 t] += ||p||^2, stats[2 t + 1] += ||u||^2 over tensor t  (the caller zeroes `stats` first)
 *   phase 1:  p -= lr * r_t * u with r_t = ||p|| / ||u|| if weight_decay != 0 and both norms > 0, else 1 (timm:
 *             always_adapt = False);  bf16 refresh + teacher / --model-ema EMA as in gv_adamw_ema.
 * `blocks`: device int32 [n_blocks][3] = {tensor, lo, hi}: element range [lo, hi) of the arena range handled by one
 * workgroup, never crossing a tensor (built once by the host from the arena layout; lo, hi multiples of 4).            */
typedef struct {
    float* p; const float* grad; float* m; float* v; void* p_bf16;
    float* teacher; void* teacher_bf16;
    const int32_t* blocks; int32_t n_blocks;
    float* stats;                /* f32 [2 * n_tensors]                                                              */
    float lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2;
    float grad_scale, clip_norm, max_grad_norm; const float* gnorm_sq;     /* gnorm_sq: sum of squares of the raw gradient */
    float teacher_momentum;
    int32_t phase;
} gv_lamb_args;
int gv_lamb(const gv_lamb_args* a, void* stream);

/* ---- fp32 operand mode (SURVEY 8d fp32 column; the reference's default arithmetic, train.py without --amp) -------
 * The same entry points with EVERY bf16 buffer of the argument struct read / written as f32 instead (activations,
 * GEMM operands incl. the weights, aux_in / aux_out, attention q/k/v/o/dO/dqkv, LayerNorm y / dy / gb, patch rows);
 * f32 fields keep their meaning.  Linear: v_mfma_f32_16x16x4_f32, any M / N / K, exact erf GELU and GELU', c_is_f32
 * must be 1, workspace ignored (pure-ACCUM launches with few output tiles split K through f32 atomics).
 * Attention: N <= 260.  A parity mode: it holds the 1e-4 logits / loss and 1e-3 gradient-norm gates against the fp32
 * oracle (tests/test_fp32_gpu.py); the training path stays bf16.                                                   */
int gv_linear_f32(const gv_linear_args* a, void* stream);
int gv_attention_fwd_f32(const gv_attention_fwd_args* a, void* stream);
int gv_attention_bwd_f32(const gv_attention_bwd_args* a, void* stream);
int gv_layernorm_fwd_f32(const gv_layernorm_fwd_args* a, void* stream);
int gv_layernorm_bwd_f32(const gv_layernorm_bwd_args* a, void* stream);
int gv_patchify_f32(const gv_patchify_args* a, void* stream);
int gv_tokens_bwd_f32(const gv_tokens_bwd_args* a, void* stream);
/* DINO head: zn / dz (l2norm), the weight-normalised last-layer matrix, and the student-logit gradient as f32 */
int gv_l2norm_fwd_f32(const gv_l2norm_fwd_args* a, void* stream);
int gv_l2norm_bwd_f32(const gv_l2norm_bwd_args* a, void* stream);
int gv_weightnorm_fwd_f32(const gv_weightnorm_fwd_args* a, void* stream);
int gv_dino_loss_f32(const gv_dino_loss_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GIPVIT_H */
