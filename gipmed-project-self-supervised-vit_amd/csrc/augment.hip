// Tile augmentation on the device -- gv_augment (include/gipvit.h).
//
// Replaces the reference's CPU augmentation of a tile (transformations.py:131-197: torchvision ColorJitter on PIL images,
// GaussianBlur(3), MyGaussianNoiseTransform 71-88, RandomVertical/HorizontalFlip, MyRotation 48-56, RandomAffine(scale),
// Cutout 10-45) for uint8 NHWC tiles that are already resident in HBM.  The random draws are made on the host
// (gipvit/augment.py) and travel as one 80-byte parameter record per tile; the arithmetic of every operation follows
// oracle/augment_oracle.py operation by operation (float32, no FMA contraction, PIL's truncations / roundings), so the
// output is byte-identical to the oracle's.
//
// HBM-bound by design: 196 608 B in + 196 608 B out per 256-px tile; one thread per output pixel computes
//   out[y, x] = cutout( noise( blur( colour( in[ geo^-1 (y, x) ] ) ) ) )
// i.e. the geometric part (dihedral flip / rot90 element + NEAREST zoom) is applied as a GATHER on the source index, the
// colour chain is evaluated on the gathered pixel (on its 3x3 neighbourhood when the blur is on).  ColorJitter's contrast
// blends with the MEAN grey level of the image as it is when the operation runs (after the operations drawn before it): a
// first kernel takes that mean per tile (integer sum of PIL's "L" values, exact).
#include "gv_common.h"

#pragma clang fp contract(off)

namespace {

struct Px { int r, g, b; };

__device__ __forceinline__ int lum(const Px p) { return (p.r * 19595 + p.g * 38470 + p.b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int blend1(float deg, int c, float f) {
    float t = deg + f * ((float)c - deg);
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (int)t;                                           // truncation (PIL casts to UINT8)
}

__device__ __forceinline__ Px hue_shift(const Px p, int shift) {
    const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
    int h = 0, s = 0;
    const int v = maxc;
    if (maxc != minc) {
        const float cr = (float)(maxc - minc);
        const float sf = cr / (float)maxc;
        const float rc = (float)(maxc - p.r) / cr, gc = (float)(maxc - p.g) / cr, bc = (float)(maxc - p.b) / cr;
        float hf = p.r == maxc ? bc - gc : (p.g == maxc ? (2.0f + rc) - bc : (4.0f + gc) - rc);
        const float t = hf / 6.0f + 1.0f;
        hf = t - truncf(t);
        h = min(max((int)(hf * 255.0f), 0), 255);
        s = min(max((int)(sf * 255.0f), 0), 255);
    }
    h = (h + shift) & 255;
    if (s == 0) return Px{v, v, v};
    const float hh = (float)h * 6.0f / 255.0f;
    const float fi = floorf(hh);
    const float f = hh - fi;
    const float fs = (float)s / 255.0f, vv = (float)v;
    const int pq = (int)floorf(vv * (1.0f - fs) + 0.5f);
    const int q = (int)floorf(vv * (1.0f - fs * f) + 0.5f);
    const int tt = (int)floorf(vv * (1.0f - fs * (1.0f - f)) + 0.5f);
    Px o;
    switch ((int)fi % 6) {
        case 0: o = Px{v, tt, pq}; break;
        case 1: o = Px{q, v, pq}; break;
        case 2: o = Px{pq, v, tt}; break;
        case 3: o = Px{pq, q, v}; break;
        case 4: o = Px{tt, pq, v}; break;
        default: o = Px{v, pq, q}; break;
    }
    o.r = min(max(o.r, 0), 255); o.g = min(max(o.g, 0), 255); o.b = min(max(o.b, 0), 255);
    return o;
}

// colour operations [first, last) of the tile's (gv_augment_params) or view's (gv_view_params) order; `mean` = the contrast
// operation's grey mean
template <class Q>
__device__ __forceinline__ Px colour(Px p, const Q& q, int first, int last, int mean) {
    for (int k = first; k < last; ++k) {
        const int op = q.order[k];
        if (op == 0) { p = Px{blend1(0.f, p.r, q.bf), blend1(0.f, p.g, q.bf), blend1(0.f, p.b, q.bf)}; }
        else if (op == 1) { const float m = (float)mean; p = Px{blend1(m, p.r, q.cf), blend1(m, p.g, q.cf), blend1(m, p.b, q.cf)}; }
        else if (op == 2) { const float l = (float)lum(p); p = Px{blend1(l, p.r, q.sf), blend1(l, p.g, q.sf), blend1(l, p.b, q.sf)}; }
        else p = hue_shift(p, q.hue);
    }
    return p;
}

__device__ __forceinline__ Px load_px(const uint8_t* t, int y, int x, int W) {
    const uint8_t* s = t + ((long)y * W + x) * 3;
    return Px{s[0], s[1], s[2]};
}

// per-tile sum of "L" over the image as it is just before the contrast operation
__global__ __launch_bounds__(256) void augment_stats_kernel(gv_augment_args a) {
    const int tile = blockIdx.y;
    const gv_augment_params q = a.params[tile];
    int kc = -1;
    for (int k = 0; k < q.n_color; ++k) if (q.order[k] == 1) kc = k;
    if (kc < 0) return;
    const uint8_t* t = a.tiles + (long)tile * a.H * a.W * 3;
    const int npx = a.H * a.W;
    unsigned long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const Px p = colour(Px{t[3 * i], t[3 * i + 1], t[3 * i + 2]}, q, 0, kc, 0);
        s += (unsigned long long)lum(p);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.stats + tile, red[0] + red[1] + red[2] + red[3]);
}

__global__ void augment_zero_kernel(unsigned long long* stats, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stats[i] = 0ull;
}

__global__ __launch_bounds__(256) void augment_kernel(gv_augment_args a) {
    const int tile = blockIdx.y;
    const int H = a.H, W = a.W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const gv_augment_params q = a.params[tile];
    uint8_t* dst = a.out + ((long)tile * H * W + i) * 3;
    if (y >= q.cut[0] && y < q.cut[1] && x >= q.cut[2] && x < q.cut[3]) { dst[0] = 0; dst[1] = 0; dst[2] = 0; return; }
    // ---- inverse geometry: zoom (NEAREST, 16.16 fixed point), then the dihedral element
    int u = y, v = x;
    if (q.zoom) {
        u = (int)(((long)q.a2 + (long)q.a0 * y) >> 16);
        v = (int)(((long)q.a2 + (long)q.a0 * x) >> 16);
        if (u < 0 || u >= H || v < 0 || v >= W) { dst[0] = 0; dst[1] = 0; dst[2] = 0; return; }
    }
    if (q.d4 & 1) { const int t = u; u = v; v = t; }
    if (q.d4 & 2) u = H - 1 - u;
    if (q.d4 & 4) v = W - 1 - v;
    const uint8_t* src = a.tiles + (long)tile * H * W * 3;
    const int mean = (int)((double)a.stats[tile] / (double)(H * W) + 0.5);
    Px p;
    if (q.blur) {
        float acc[3] = {0.f, 0.f, 0.f};
        const float k1[3] = {q.ks, q.kc, q.ks};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                int yy = u + dy - 1, xx = v + dx - 1;
                yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);       // reflect padding
                xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                const Px c = colour(load_px(src, yy, xx, W), q, 0, q.n_color, mean);
                const float w = k1[dy] * k1[dx];
                acc[0] = acc[0] + w * (float)c.r; acc[1] = acc[1] + w * (float)c.g; acc[2] = acc[2] + w * (float)c.b;
            }
        p = Px{(int)fminf(fmaxf(rintf(acc[0]), 0.f), 255.f), (int)fminf(fmaxf(rintf(acc[1]), 0.f), 255.f), (int)fminf(fmaxf(rintf(acc[2]), 0.f), 255.f)};
    } else {
        p = colour(load_px(src, u, v, W), q, 0, q.n_color, mean);
    }
    if (q.sigma > 0.f) {
        int c3[3] = {p.r, p.g, p.b};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t h = q.seed + 0x9E3779B9u * (uint32_t)((u * W + v) * 3 + c + 1);
            h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
            const float z = a.ztable[h >> 22];
            float xf = (float)c3[c] / 255.0f;
            xf = xf + q.sigma * z;
            xf = fminf(fmaxf(xf, 0.0f), 1.0f);
            c3[c] = (int)(255.0f * xf);
        }
        p = Px{c3[0], c3[1], c3[2]};
    }
    dst[0] = (uint8_t)p.r; dst[1] = (uint8_t)p.g; dst[2] = (uint8_t)p.b;
}

// ---------------------------------------------------------------------------------------------------------------------
// gv_crop_augment: DINO views.  view[y, x] = solarise( blur3( gray( jitter( resized_crop[y, x] ) ) ) ), the resized crop taken
// straight from the tile (gv_crop_resize's float32 bilinear arithmetic, byte for byte) -- it is never stored.
// ---------------------------------------------------------------------------------------------------------------------
struct Box { const uint8_t* src; int y0, x0, h, w, flip, W, out; float sy, sx; };

__device__ __forceinline__ Box load_box(const gv_crop_augment_args& a, int n) {
    const int* bx = a.boxes + n * 6;
    Box b;
    b.src = a.tiles + (long)bx[0] * a.tile_h * a.tile_w * 3;
    b.y0 = bx[1]; b.x0 = bx[2]; b.h = bx[3]; b.w = bx[4]; b.flip = bx[5]; b.W = a.tile_w; b.out = a.out_size;
    b.sy = (float)b.h / (float)b.out; b.sx = (float)b.w / (float)b.out;
    return b;
}

// pixel (oy, oxo) of the resized (and flipped) crop
__device__ __forceinline__ Px resample(const Box& b, int oy, int oxo) {
    float fy = b.sy * ((float)oy + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int iy0 = (int)fy, iy1 = iy0 + (iy0 < b.h - 1 ? 1 : 0);
    const float ly = fy - (float)iy0, ly0 = 1.0f - ly;
    const int ox = b.flip ? b.out - 1 - oxo : oxo;
    float fx = b.sx * ((float)ox + 0.5f) - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    const int ix0 = (int)fx, ix1 = ix0 + (ix0 < b.w - 1 ? 1 : 0);
    const float lx = fx - (float)ix0, lx0 = 1.0f - lx;
    const uint8_t* r0 = b.src + ((long)(b.y0 + iy0) * b.W + b.x0) * 3;
    const uint8_t* r1 = b.src + ((long)(b.y0 + iy1) * b.W + b.x0) * 3;
    int c3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = lx0 * (float)r0[ix0 * 3 + c] + lx * (float)r0[ix1 * 3 + c];
        const float bot = lx0 * (float)r1[ix0 * 3 + c] + lx * (float)r1[ix1 * 3 + c];
        float v = __builtin_rintf(ly0 * top + ly * bot);
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        c3[c] = (int)v;
    }
    return Px{c3[0], c3[1], c3[2]};
}

__device__ __forceinline__ Px jitter_gray(Px p, const gv_view_params& q, int mean) {
    p = colour(p, q, 0, q.n_color, mean);
    if (q.gray) { const int l = lum(p); p = Px{l, l, l}; }
    return p;
}

// per-crop sum of "L" over the view as it is just before the contrast operation
__global__ __launch_bounds__(256) void view_stats_kernel(gv_crop_augment_args a) {
    const int n = blockIdx.y;
    const gv_view_params q = a.params[n];
    int kc = -1;
    for (int k = 0; k < q.n_color; ++k) if (q.order[k] == 1) kc = k;
    if (kc < 0) return;
    const Box b = load_box(a, n);
    const int npx = a.out_size * a.out_size;
    unsigned long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int y = i / a.out_size, x = i - y * a.out_size;
        s += (unsigned long long)lum(colour(resample(b, y, x), q, 0, kc, 0));
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)a.stats + n, red[0] + red[1] + red[2] + red[3]);
}

// One 256-thread block per 32 x 32 output region.  Phase 1: the resized crop's pixels of the region (+ a one-pixel halo when the
// view is blurred) go through the colour chain ONCE each and are parked in LDS (packed rgb); phase 2: every thread finishes
// 4 adjacent pixels of a row -- the 3x3 taps summed in the oracle's raster order, solarise -- and stores 12 bytes as three
// dwords.  (Evaluating the chain per tap cost 9 evaluations per blurred pixel: +2.5 ms per B = 64 step.)
constexpr int VT = 32;
__global__ __launch_bounds__(256) void crop_augment_kernel(gv_crop_augment_args a) {
    __shared__ uint32_t win[(VT + 2) * (VT + 2)];
    const int out = a.out_size;
    const int n = blockIdx.z, ty0 = blockIdx.y * VT, tx0 = blockIdx.x * VT;
    const gv_view_params q = a.params[n];
    const Box b = load_box(a, n);
    const int mean = (int)((double)a.stats[n] / (double)(out * out) + 0.5);
    const int halo = q.blur ? 1 : 0, side = VT + 2 * halo;
#pragma unroll 1
    for (int i = threadIdx.x; i < side * side; i += 256) {
        const int wy = i / side, wx = i - wy * side;
        int yy = ty0 + wy - halo, xx = tx0 + wx - halo;
        yy = yy < 0 ? -yy : (yy >= out ? 2 * out - 2 - yy : yy);                   // reflect padding (positions past a ragged region's
        xx = xx < 0 ? -xx : (xx >= out ? 2 * out - 2 - xx : xx);                   //  edge stay inside the crop: never stored)
        yy = yy < 0 ? 0 : yy; xx = xx < 0 ? 0 : xx;
        const Px c = jitter_gray(resample(b, yy, xx), q, mean);
        win[i] = (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
    }
    __syncthreads();
    const int ry = threadIdx.x >> 3, rx = (threadIdx.x & 7) * 4;                  // this thread's row and first column inside the region
    const int oy = ty0 + ry, ox = tx0 + rx;
    if (oy >= out || ox >= out) return;
    uint32_t pk[3] = {0u, 0u, 0u};
    const float k1[3] = {q.ks, q.kc, q.ks};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c3[3];
        if (q.blur) {
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint32_t c = win[(ry + dy) * side + rx + i + dx];
                    const float w = k1[dy] * k1[dx];
                    acc[0] = acc[0] + w * (float)(c & 255u); acc[1] = acc[1] + w * (float)((c >> 8) & 255u); acc[2] = acc[2] + w * (float)((c >> 16) & 255u);
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) c3[c] = (int)fminf(fmaxf(rintf(acc[c]), 0.f), 255.f);
        } else {
            const uint32_t c = win[ry * side + rx + i];
            c3[0] = (int)(c & 255u); c3[1] = (int)((c >> 8) & 255u); c3[2] = (int)((c >> 16) & 255u);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = (q.solar >= 0 && c3[c] >= q.solar) ? 255 - c3[c] : c3[c];
            const int bb = i * 3 + c;
            pk[bb >> 2] |= (uint32_t)v << (8 * (bb & 3));
        }
    }
    uint32_t* dst = (uint32_t*)(a.out + ((long)n * out * out + (long)oy * out + ox) * 3);
    dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2];
}

}  // namespace

extern "C" int gv_crop_augment(const gv_crop_augment_args* a, void* stream) {
    GV_REQUIRE(a && a->tiles && a->out && a->boxes && a->params && a->stats, GV_E_NULL, "gv_crop_augment: null pointer");
    GV_REQUIRE(a->n_crops > 0 && a->n_tiles > 0 && a->tile_h > 0 && a->tile_w > 0, GV_E_SHAPE, "gv_crop_augment: bad shape");
    GV_REQUIRE(a->out_size > 1 && a->out_size % 4 == 0, GV_E_SHAPE, "gv_crop_augment: out_size=%d must be a multiple of 4 (> 1)", a->out_size);
    GV_REQUIRE(a->n_crops <= 65535, GV_E_SHAPE, "gv_crop_augment: at most 65535 crops per call (got %d)", a->n_crops);
    GV_REQUIRE(gv_aligned(a->out, 4) && gv_aligned(a->boxes, 4) && gv_aligned(a->params, 4) && gv_aligned(a->stats, 8), GV_E_ALIGN,
               "gv_crop_augment: out / boxes / params / stats misaligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(augment_zero_kernel, dim3((a->n_crops + 255) / 256), dim3(256), 0, s, (unsigned long long*)a->stats, a->n_crops);
    hipLaunchKernelGGL(view_stats_kernel, dim3(8, a->n_crops), dim3(256), 0, s, *a);
    GV_LAUNCH_CHECK("gv_crop_augment(stats)");
    const int nt = (a->out_size + VT - 1) / VT;
    hipLaunchKernelGGL(crop_augment_kernel, dim3(nt, nt, a->n_crops), dim3(256), 0, s, *a);
    GV_LAUNCH_CHECK("gv_crop_augment");
    return GV_OK;
}

extern "C" int gv_augment(const gv_augment_args* a, void* stream) {
    GV_REQUIRE(a && a->tiles && a->out && a->params && a->stats && a->ztable, GV_E_NULL, "gv_augment: null pointer");
    GV_REQUIRE(a->n > 0 && a->H > 1 && a->W > 1 && a->H == a->W, GV_E_SHAPE, "gv_augment: need n > 0 square tiles (got %d x %d x %d)", a->n, a->H, a->W);
    GV_REQUIRE(a->tiles != a->out, GV_E_UNSUPPORTED, "gv_augment: in-place operation is not supported (the geometry is a gather)");
    hipStream_t s = (hipStream_t)stream;
    // zeroed by a kernel, not hipMemsetAsync: with an H2D copy in flight on another stream (the tile prefetcher) the memset
    // stalled the launch loop for ~5 ms per step (tools/handover_bench.py)
    hipLaunchKernelGGL(augment_zero_kernel, dim3((a->n + 255) / 256), dim3(256), 0, s, (unsigned long long*)a->stats, a->n);
    const int npx = a->H * a->W;
    hipLaunchKernelGGL(augment_stats_kernel, dim3(32, a->n), dim3(256), 0, s, *a);
    GV_LAUNCH_CHECK("gv_augment(stats)");
    hipLaunchKernelGGL(augment_kernel, dim3((npx + 255) / 256, a->n), dim3(256), 0, s, *a);
    GV_LAUNCH_CHECK("gv_augment");
    return GV_OK;
}
