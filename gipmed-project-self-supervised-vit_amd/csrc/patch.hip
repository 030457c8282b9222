// patchify: NHWC uint8 tile crop windows -> normalised bf16 patch rows.
// Replaces the im2col half of PatchEmbed's Conv2d(3, D, 16, 16) (vit.pyc@L167-170)
// fused with the reference's ToTensor + Normalize (transformations.py:124-128) and
// the crop slicing of the multi-crop input contract.
//
// HBM-bound.  One thread per (image, patch, pixel row): reads the 48 contiguous
// bytes of 16 RGB pixels and writes three 32-B runs (one per channel) so that a
// patch row is laid out k = c*256 + py*16 + px, the flatten order of the conv
// weight [D, 3, 16, 16].  A wave covers 4 horizontally adjacent patches x 16 pixel
// rows: reads are 192-B contiguous per pixel row, writes 512-B contiguous per
// (patch, channel).   algorithmic bytes / image: crop^2 * 3 in + (crop/16)^2 * 1536 out.
#include "gv_common.h"

namespace {

template <bool F32OUT>      // fp32 parity mode: f32 patch rows (f32path.hip)
__global__ __launch_bounds__(256) void patchify_kernel(gv_patchify_args a, int P, int side, float s0, float s1, float s2,
                                                       float o0, float o1, float o2) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.n_img * P * 16;
    if (t >= total) return;
    const int py = (int)(t & 15);
    const long ip = t >> 4;
    const int patch = (int)(ip % P);
    const int img = (int)(ip / P);
    const int tile = img % a.n_tiles, win = img / a.n_tiles;
    const int prow = patch / side, pcol = patch - prow * side;
    const int y = a.win_y[win] + prow * 16 + py, x = a.win_x[win] + pcol * 16;
    const uint8_t* src = a.tiles + (long)tile * a.img_stride + ((long)y * a.tile_w + x) * 3;

    // 48 bytes at arbitrary alignment -> 12 aligned dwords (+ up to 3 tail bytes)
    const uintptr_t addr = (uintptr_t)src;
    const uint32_t* w = (const uint32_t*)(addr & ~(uintptr_t)3);
    const int mis = (int)(addr & 3);
    uint32_t d[13];
#pragma unroll
    for (int i = 0; i < 12; ++i) d[i] = w[i];
    d[12] = 0;
    if (mis) {
        const uint8_t* tail = (const uint8_t*)(w + 12);
        for (int i = 0; i < mis; ++i) d[12] |= (uint32_t)tail[i] << (8 * i);
    }
    uint32_t u[12];
    if (mis == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = d[i];
    } else {
        const int sh = mis * 8;
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = (d[i] >> sh) | (d[i + 1] << (32 - sh));
    }
    // byte j of the run = pixel j/3, channel j%3
    float px[3][16];
#pragma unroll
    for (int j = 0; j < 48; ++j) {
        const float v = (float)((u[j >> 2] >> (8 * (j & 3))) & 0xFF);
        const int c = j % 3;
        px[c][j / 3] = v * (c == 0 ? s0 : c == 1 ? s1 : s2) + (c == 0 ? o0 : c == 1 ? o1 : o2);
    }
    if (a.fill) {
        const float* f = a.fill + (long)tile * 8;
        if (f[7] != 0.f && (float)y >= f[0] && (float)y < f[1]) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((float)(x + i) >= f[2] && (float)(x + i) < f[3]) { px[0][i] = f[4]; px[1][i] = f[5]; px[2][i] = f[6]; }
        }
    }
    if constexpr (F32OUT) {
        float* out = (float*)a.patches + ip * 768 + py * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; i += 4) *(f32x4*)(out + c * 256 + i) = f32x4{px[c][i], px[c][i + 1], px[c][i + 2], px[c][i + 3]};
    } else {
    bf16* out = (bf16*)a.patches + ip * 768 + py * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bf16x8 lo, hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = (bf16)px[c][i]; hi[i] = (bf16)px[c][8 + i]; }
        *(bf16x8*)(out + c * 256) = lo;
        *(bf16x8*)(out + c * 256 + 8) = hi;
    }
    }
}

// ---- random-resized-crop (+ horizontal flip) of NHWC u8 tiles: the DINO multi-crop input
// stage.  Semantics = torchvision's tensor-mode resized_crop with
// antialias off: the box (y0, x0, h, w) of the tile is resampled to out x out with
// F.interpolate(float32, mode='bilinear', align_corners=False), rounded half-to-even and
// clamped to u8; flip mirrors the OUTPUT columns.  One thread per 4 output pixels (12
// contiguous bytes); float32 arithmetic in the oracle's association, no FMA contraction,
// so the bytes are reproducible against the CPU restatement.
__global__ __launch_bounds__(256) void crop_resize_kernel(gv_crop_resize_args a) {
#pragma clang fp contract(off)
    const int out = a.out_size, q = out >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.n_crops * out * q;
    if (t >= total) return;
    const int xq = (int)(t % q);
    const int oy = (int)((t / q) % out);
    const int n = (int)(t / ((long)q * out));
    const int* bx = a.boxes + n * 6;
    const int tile = bx[0], y0 = bx[1], x0 = bx[2], h = bx[3], w = bx[4], flip = bx[5];
    const uint8_t* src = a.tiles + (long)tile * a.tile_h * a.tile_w * 3;
    const float sy = (float)h / (float)out, sx = (float)w / (float)out;
    float fy = sy * ((float)oy + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int iy0 = (int)fy, iy1 = iy0 + (iy0 < h - 1 ? 1 : 0);
    const float ly = fy - (float)iy0, ly0 = 1.0f - ly;
    const uint8_t* r0 = src + ((long)(y0 + iy0) * a.tile_w + x0) * 3;
    const uint8_t* r1 = src + ((long)(y0 + iy1) * a.tile_w + x0) * 3;
    uint32_t pk[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oxo = xq * 4 + i;                       // output column
        const int ox = flip ? out - 1 - oxo : oxo;        // column of the un-flipped resample
        float fx = sx * ((float)ox + 0.5f) - 0.5f;
        fx = fx < 0.f ? 0.f : fx;
        const int ix0 = (int)fx, ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
        const float lx = fx - (float)ix0, lx0 = 1.0f - lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = lx0 * (float)r0[ix0 * 3 + c] + lx * (float)r0[ix1 * 3 + c];
            const float bot = lx0 * (float)r1[ix0 * 3 + c] + lx * (float)r1[ix1 * 3 + c];
            float v = __builtin_rintf(ly0 * top + ly * bot);
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            const int b = i * 3 + c;
            pk[b >> 2] |= (uint32_t)v << (8 * (b & 3));
        }
    }
    uint32_t* dst = (uint32_t*)(a.out + ((long)n * out * out + (long)oy * out + xq * 4) * 3);
    dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2];
}

}  // namespace

extern "C" int gv_crop_resize(const gv_crop_resize_args* a, void* stream) {
    GV_REQUIRE(a && a->tiles && a->out && a->boxes, GV_E_NULL, "gv_crop_resize: null pointer");
    GV_REQUIRE(a->n_crops > 0 && a->n_tiles > 0 && a->tile_h > 0 && a->tile_w > 0, GV_E_SHAPE, "gv_crop_resize: bad shape");
    GV_REQUIRE(a->out_size > 0 && a->out_size % 4 == 0, GV_E_SHAPE, "gv_crop_resize: out_size=%d must be a positive multiple of 4", a->out_size);
    GV_REQUIRE(gv_aligned(a->out, 4) && gv_aligned(a->boxes, 4), GV_E_ALIGN, "gv_crop_resize: out / boxes must be 4-byte aligned");
    const long total = (long)a->n_crops * a->out_size * (a->out_size / 4);
    hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_crop_resize");
    return GV_OK;
}

template <bool F32OUT> static int patchify_launch(const gv_patchify_args* a, void* stream) {
    GV_REQUIRE(a && a->tiles && a->patches, GV_E_NULL, "gv_patchify: null pointer");
    GV_REQUIRE(a->crop > 0 && a->crop % 16 == 0, GV_E_SHAPE, "gv_patchify: crop=%d must be a positive multiple of 16", a->crop);
    GV_REQUIRE(a->n_win >= 1 && a->n_win <= 16 && a->n_tiles >= 1 && a->n_img == a->n_win * a->n_tiles, GV_E_SHAPE,
               "gv_patchify: n_img (%d) must equal n_win (%d) * n_tiles (%d), n_win <= 16", a->n_img, a->n_win, a->n_tiles);
    for (int w = 0; w < a->n_win; ++w)
        GV_REQUIRE(a->win_y[w] >= 0 && a->win_x[w] >= 0 && a->win_y[w] + a->crop <= a->tile_h && a->win_x[w] + a->crop <= a->tile_w,
                   GV_E_SHAPE, "gv_patchify: window %d (%d,%d)+%d leaves the %dx%d tile", w, a->win_y[w], a->win_x[w], a->crop, a->tile_h, a->tile_w);
    GV_REQUIRE(gv_aligned(a->patches, 16), GV_E_ALIGN, "gv_patchify: patches must be 16-byte aligned");
    for (int c = 0; c < 3; ++c) GV_REQUIRE(a->std[c] > 0.f, GV_E_SHAPE, "gv_patchify: std must be > 0");
    const int side = a->crop / 16, P = side * side;
    const long total = (long)a->n_img * P * 16;
    const float s0 = 1.0f / (255.0f * a->std[0]), s1 = 1.0f / (255.0f * a->std[1]), s2 = 1.0f / (255.0f * a->std[2]);
    const float o0 = -a->mean[0] / a->std[0], o1 = -a->mean[1] / a->std[1], o2 = -a->mean[2] / a->std[2];
    hipLaunchKernelGGL(patchify_kernel<F32OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a, P, side,
                       s0, s1, s2, o0, o1, o2);
    GV_LAUNCH_CHECK("gv_patchify");
    return GV_OK;
}
extern "C" int gv_patchify(const gv_patchify_args* a, void* stream) { return patchify_launch<false>(a, stream); }
extern "C" int gv_patchify_f32(const gv_patchify_args* a, void* stream) { return patchify_launch<true>(a, stream); }
