// a3/a4: heatmap argmax + zero-padded 3x3 window + bounded Gaussian fit (HBM-bound streaming kernel).
// Reference: balldetection/helper_balldetection.py:29-110, tabledetection/helper_tabledetection.py:50-156.
//
// Kernel 1 (argmax_partial): grid (nblk, n_maps), 256 threads; each workgroup streams one contiguous
//   slice of a heatmap with 16-byte loads (4 in flight per lane), keeps (max, first index) per lane,
//   reduces across the wave with DPP shuffles and across the 4 waves through LDS, and writes one partial.
//   Algorithmic traffic = H*W*4 bytes per heatmap, read exactly once.
// Kernel 2 (argmax_finish): one wave per map reduces the partials (ties -> smaller index, NaN wins like
//   torch.argmax), gathers the zero-padded window and stores index + window.
// Kernel 3 (fit): one lane per map runs the L-BFGS-B fit of lbfgsb.h in fp64 and rescales to image pixels.
#include "no_packed_fp32_begin.h"      // this unit's kernels run beside the CNN's chain kernels: no packed fp32 (common.h)
#include "common.h"


#include "lbfgsb.h"

namespace ttup {

struct Best { float v; long long i; };

__device__ __forceinline__ bool better(float v, long long i, float bv, long long bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_down(b.v, off, 64);
        const long long oi = __shfl_down(b.i, off, 64);
        if (better(ov, oi, b.v, b.i)) { b.v = ov; b.i = oi; }
    }
    return b;
}

template <bool VEC>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ heat, long long hw, int nblk,
                                                             float* __restrict__ pv, long long* __restrict__ pi) {
    const int map = ttup_bid_y(), blk = ttup_bid_x(), tid = ttup_tid_x();
    const float* h = heat + (size_t)map * hw;
    // slice boundaries in units of 4 floats so that vector loads stay aligned
    const long long quads = (hw + 3) / 4;
    const long long q0 = quads * blk / nblk, q1 = quads * (blk + 1) / nblk;
    Best b; b.v = -INFINITY; b.i = 0x7fffffffffffffffLL;
    if (VEC) {
        const float4* h4 = (const float4*)h;
        long long q = q0 + tid;
        for (; q + 3 * 256 < q1; q += 4 * 256) {
            const float4 a0 = h4[q], a1 = h4[q + 256], a2 = h4[q + 512], a3 = h4[q + 768];
            const float4 arr[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long base = (q + k * 256) * 4;
                if (better(arr[k].x, base + 0, b.v, b.i)) { b.v = arr[k].x; b.i = base + 0; }
                if (better(arr[k].y, base + 1, b.v, b.i)) { b.v = arr[k].y; b.i = base + 1; }
                if (better(arr[k].z, base + 2, b.v, b.i)) { b.v = arr[k].z; b.i = base + 2; }
                if (better(arr[k].w, base + 3, b.v, b.i)) { b.v = arr[k].w; b.i = base + 3; }
            }
        }
        for (; q < q1; q += 256) {
            const float4 a = h4[q];
            const long long base = q * 4;
            if (better(a.x, base + 0, b.v, b.i)) { b.v = a.x; b.i = base + 0; }
            if (better(a.y, base + 1, b.v, b.i)) { b.v = a.y; b.i = base + 1; }
            if (better(a.z, base + 2, b.v, b.i)) { b.v = a.z; b.i = base + 2; }
            if (better(a.w, base + 3, b.v, b.i)) { b.v = a.w; b.i = base + 3; }
        }
    } else {
        const long long e1 = q1 * 4 < hw ? q1 * 4 : hw;
        for (long long e = q0 * 4 + tid; e < e1; e += 256) {
            const float v = h[e];
            if (better(v, e, b.v, b.i)) { b.v = v; b.i = e; }
        }
    }
    b = wave_best(b);
    __shared__ float sv[4];
    __shared__ long long si[4];
    if ((tid & 63) == 0) { sv[tid >> 6] = b.v; si[tid >> 6] = b.i; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k) if (better(sv[k], si[k], b.v, b.i)) { b.v = sv[k]; b.i = si[k]; }
        pv[(size_t)map * nblk + blk] = b.v;
        pi[(size_t)map * nblk + blk] = b.i;
    }
}

__global__ __launch_bounds__(64) void argmax_finish_kernel(const float* __restrict__ heat, int H, int W, int nblk,
                                                           const float* __restrict__ pv, const long long* __restrict__ pi,
                                                           long long* __restrict__ argmax, float* __restrict__ win) {
    const int map = ttup_bid_x(), lane = ttup_tid_x();
    Best b; b.v = -INFINITY; b.i = 0x7fffffffffffffffLL;
    for (int k = lane; k < nblk; k += 64) {
        const float v = pv[(size_t)map * nblk + k];
        const long long i = pi[(size_t)map * nblk + k];
        if (better(v, i, b.v, b.i)) { b.v = v; b.i = i; }
    }
    b = wave_best(b);
    const long long idx = __shfl(b.i, 0, 64);
    if (lane == 0) argmax[map] = idx;
    if (lane < 9) {
        const int y = (int)(idx / W) + lane / 3 - 1, x = (int)(idx % W) + lane % 3 - 1;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = heat[(size_t)map * H * W + (size_t)y * W + x];
        win[(size_t)map * 9 + lane] = v;
    }
}

__global__ void fit_kernel(const long long* __restrict__ argmax, const float* __restrict__ win, int n_maps, int H, int W,
                           double scale_x, double scale_y, int variant, double* __restrict__ out) {
    const int i = ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= n_maps) return;
    float w[9];
    for (int k = 0; k < 9; ++k) w[k] = win[(size_t)i * 9 + k];
    double xo, yo;
    refine_window(w, variant, &xo, &yo, nullptr);
    const long long idx = argmax[i];
    // index goes through float32 in the reference (x_max[b].float(), helper_balldetection.py:96)
    const double xs = (double)(float)(idx % W) - 1.0 + xo;
    const double ys = (double)(float)(idx / W) - 1.0 + yo;
    out[(size_t)i * 3 + 0] = (xs + 0.5) * scale_x - 0.5;
    out[(size_t)i * 3 + 1] = (ys + 0.5) * scale_y - 0.5;
    out[(size_t)i * 3 + 2] = 1.0;      // visibility: always 1 (helper_balldetection.py:13,82 / helper_tabledetection.py:142)
}

// ---- fused tail of the CNN: y = relu(base + sum up(t_k)) (stage-4 fuse output 0, wasb.py:236-243), heat = w . bf16(y) + b
// (final_layers[0] channel 1, wasb.py:484,606) and the per-workgroup argmax partial, in one pass over the pixels.
// y is rounded to bf16 before the dot product exactly as the unfused path stores it, so heatmaps are bit-identical.
struct UpsumHeadArgs {
    const bf16_t* base; const bf16_t* t[3]; int shift[3]; int n;
    const float* w; float bias;
    float* heat; float* pv; long long* pi;
    int H, W, nblk;
};

__device__ __forceinline__ void add8(float* v, const bf16_t* p) {
    const uint4 r = *(const uint4*)p;
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32((bf16_t)(w[k] & 0xffff)); v[2 * k + 1] += bf16_to_f32((bf16_t)(w[k] >> 16)); }
}

__global__ __launch_bounds__(256) void upsum_head_kernel(UpsumHeadArgs a) {
    const int map = ttup_bid_y(), blk = ttup_bid_x(), tid = ttup_tid_x();
    const long long hw = (long long)a.H * a.W;
    const long long e = (long long)blk * 256 + tid;
    Best b; b.v = -INFINITY; b.i = 0x7fffffffffffffffLL;
    if (e < hw) {
        const int y = (int)(e / a.W), x = (int)(e % a.W);
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
        const bf16_t* bp = a.base + ((size_t)map * hw + e) * 16;
        add8(v, bp); add8(v + 8, bp + 8);
        for (int t = 0; t < a.n; ++t) {
            const int sh = a.shift[t], hh = a.H >> sh, ww = a.W >> sh;
            const bf16_t* tp = a.t[t] + (((size_t)map * hh + (y >> sh)) * ww + (x >> sh)) * 16;
            add8(v, tp); add8(v + 8, tp + 8);
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float r = v[k] > 0.f ? v[k] : 0.f;
            acc = fmaf(bf16_to_f32(f32_to_bf16(r)), a.w[k], acc);
        }
        acc += a.bias;
        a.heat[(size_t)map * hw + e] = acc;
        b.v = acc; b.i = e;
    }
    b = wave_best(b);
    __shared__ float sv[4];
    __shared__ long long si[4];
    if ((tid & 63) == 0) { sv[tid >> 6] = b.v; si[tid >> 6] = b.i; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k) if (better(sv[k], si[k], b.v, b.i)) { b.v = sv[k]; b.i = si[k]; }
        a.pv[(size_t)map * a.nblk + blk] = b.v;
        a.pi[(size_t)map * a.nblk + blk] = b.i;
    }
}

// partials scratch needed by launch_upsum_head for `n_maps` heatmaps
size_t upsum_head_ws_bytes(int n_maps, int H, int W) { return (size_t)n_maps * (((size_t)H * W + 255) / 256) * 12 + 64; }

int launch_upsum_head(const void* base, const void* const* terms, const int* shifts, int n_terms, const float* w_dev, float bias,
                      float* heat, int n_maps, int H, int W, long long* argmax, float* win, void* ws, size_t ws_bytes, hipStream_t st) {
    const long long hw = (long long)H * W;
    const int nblk = (int)((hw + 255) / 256);
    TTUP_REQUIRE(ws && ws_bytes >= upsum_head_ws_bytes(n_maps, H, W), TTUP_EINVAL, "upsum_head: workspace too small");
    UpsumHeadArgs a;
    a.base = (const bf16_t*)base; a.n = n_terms;
    for (int k = 0; k < 3; ++k) { a.t[k] = k < n_terms ? (const bf16_t*)terms[k] : nullptr; a.shift[k] = k < n_terms ? shifts[k] : 0; }
    a.w = w_dev; a.bias = bias; a.heat = heat; a.H = H; a.W = W; a.nblk = nblk;
    a.pi = (long long*)ws; a.pv = (float*)(a.pi + (size_t)n_maps * nblk);
    hipLaunchKernelGGL(upsum_head_kernel, dim3(nblk, n_maps), dim3(256), 0, st, a);
    TTUP_LAUNCH_CHECK();
    if (argmax || win) {
        TTUP_REQUIRE(argmax && win, TTUP_EINVAL, "upsum_head: argmax and window outputs come together");
        hipLaunchKernelGGL(argmax_finish_kernel, dim3(n_maps), dim3(64), 0, st, heat, H, W, nblk, a.pv, a.pi, argmax, win);
        TTUP_LAUNCH_CHECK();
    }
    return TTUP_OK;
}

int launch_argmax_finish(const float* heat, int n_maps, int H, int W, int nblk, const float* pv, const long long* pi, long long* argmax, float* win, hipStream_t st) {
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(n_maps), dim3(64), 0, st, heat, H, W, nblk, pv, pi, argmax, win);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

static int pick_nblk(int n_maps, long long hw) {
    long long nblk = 2048 / (n_maps > 0 ? n_maps : 1);
    const long long cap = hw / 4096 > 0 ? hw / 4096 : 1;     // at least 16 KB per workgroup
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    return (int)nblk;
}

int refine_argmax(const float* heat, int n_maps, int H, int W, long long* argmax, float* win, void* ws, size_t ws_bytes, hipStream_t st) {
    const long long hw = (long long)H * W;
    const int nblk = pick_nblk(n_maps, hw);
    const size_t need = (size_t)n_maps * nblk * (sizeof(float) + sizeof(long long));
    TTUP_REQUIRE(ws && ws_bytes >= need, TTUP_EINVAL, "refine workspace too small: %zu < %zu", ws_bytes, need);
    long long* pi = (long long*)ws;
    float* pv = (float*)(pi + (size_t)n_maps * nblk);
    const bool vec = (hw % 4 == 0) && (((uintptr_t)heat) % 16 == 0);
    if (vec) hipLaunchKernelGGL(argmax_partial_kernel<true>, dim3(nblk, n_maps), dim3(256), 0, st, heat, hw, nblk, pv, pi);
    else hipLaunchKernelGGL(argmax_partial_kernel<false>, dim3(nblk, n_maps), dim3(256), 0, st, heat, hw, nblk, pv, pi);
    TTUP_LAUNCH_CHECK();
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(n_maps), dim3(64), 0, st, heat, H, W, nblk, pv, pi, argmax, win);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

}  // namespace ttup

using namespace ttup;

extern "C" size_t ttup_refine_workspace_bytes(int n_maps, int height, int width) {
    if (n_maps <= 0 || height <= 0 || width <= 0) return 256;
    const int nblk = pick_nblk(n_maps, (long long)height * width);
    // partials + internal argmax/window staging (used when the caller passes null outputs)
    return (size_t)n_maps * nblk * 12 + (size_t)n_maps * (8 + 36) + 256;
}

extern "C" int ttup_refine_windows(const int64_t* argmax_dev, const float* win_dev, int n_maps, int height, int width,
                                   int img_w, int img_h, int variant, double* out_xyv_dev, void* stream) {
    TTUP_REQUIRE(argmax_dev && win_dev && out_xyv_dev, TTUP_EINVAL, "ttup_refine_windows: null pointer");
    TTUP_REQUIRE(n_maps >= 0 && height > 0 && width > 0, TTUP_EINVAL, "ttup_refine_windows: bad shape");
    TTUP_REQUIRE(variant == TTUP_REFINE_BALL || variant == TTUP_REFINE_TABLE, TTUP_EINVAL, "unknown refine variant %d", variant);
    if (n_maps == 0) return TTUP_OK;
    hipLaunchKernelGGL(fit_kernel, dim3(cdiv(n_maps, 64)), dim3(64), 0, (hipStream_t)stream, (const long long*)argmax_dev, win_dev,
                       n_maps, height, width, (double)img_w / width, (double)img_h / height, variant, out_xyv_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

extern "C" int ttup_refine(const float* heat_dev, int n_maps, int height, int width, int img_w, int img_h, int variant,
                           double* out_xyv_dev, int64_t* argmax_dev, float* win_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    TTUP_REQUIRE(heat_dev, TTUP_EINVAL, "ttup_refine: null heatmap");
    TTUP_REQUIRE(n_maps >= 0 && height > 0 && width > 0, TTUP_EINVAL, "Heatmaps must have shape (B, C, H, W)");
    if (n_maps == 0) return TTUP_OK;
    const size_t need = ttup_refine_workspace_bytes(n_maps, height, width);
    TTUP_REQUIRE(ws_dev && ws_bytes >= need, TTUP_EINVAL, "ttup_refine: workspace %zu < %zu bytes", ws_bytes, need);
    const int nblk = pick_nblk(n_maps, (long long)height * width);
    char* p = (char*)ws_dev;
    const size_t part = (size_t)n_maps * nblk * 12;
    long long* am = argmax_dev ? (long long*)argmax_dev : (long long*)(p + ((part + 15) & ~(size_t)15));
    float* wn = win_dev ? win_dev : (float*)(p + ((part + 15) & ~(size_t)15) + (size_t)n_maps * 8);
    int rc = refine_argmax(heat_dev, n_maps, height, width, am, wn, ws_dev, part, (hipStream_t)stream);
    if (rc) return rc;
    if (!out_xyv_dev) return TTUP_OK;
    return ttup_refine_windows((const int64_t*)am, wn, n_maps, height, width, img_w, img_h, variant, out_xyv_dev, stream);
}

#include "no_packed_fp32_end.h"
