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

// colour operations [first, last) of the tile's order; `mean` = the contrast operation's grey mean
__device__ __forceinline__ Px colour(Px p, const gv_augment_params& q, int first, int last, int mean) {
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

}  // namespace

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
