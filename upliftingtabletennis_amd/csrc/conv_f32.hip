// fp32 convolution on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, one rounding per product, a
// k-ordered fma chain) -- the arithmetic of the parity path (TTUP_DTYPE_F32) and of the certified-argmax re-evaluation
// (csrc/certify.hip), 6-8x faster than the one-thread-per-output direct kernel it replaces (conv_direct_f32_kernel, kept in
// conv.hip and selectable with TTUP_F32_DIRECT=1 as a cross-check).
// Reference: the conv / BN(folded) / ReLU / residual call sites of balldetection/models/wasb.py:48-64, :85-105, :227-245, :446-451.
//
// GEMM view per output tile (8 rows x 16 columns):  D[cout][px] = sum_{chunk, tap, c} W[tap][c][cout] * X[c][px(tap)]
//   A operand = weights (16 couts x 4 channels per MFMA), B operand = 16 consecutive output columns of one row,
//   both read from LDS: the input halo tile is staged CHANNEL-major ([8 channels][pixels], plane stride = 16 mod 32 dwords so
//   the four channel planes of a fragment fall on disjoint banks), the weight chunk as [tap][8 channels][cout (+16 pad)].
//   A lane ends with 4 consecutive couts of one pixel -> float4 NHWC stores.
// Persistent workgroups walk the tiles; `n_active` (device memory, optional) overrides the batch so that a launch sized
// for the largest batch does only the work that a previous kernel decided on (no host synchronisation).
#include "conv.h"

namespace ttup {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvF32Args {
    const float* src0; const float* src1; const float* w; const float* bias; const float* residual; float* dst;
    int c0, c1, cout, H, W, OH, OW, relu, batch;
    const int* n_active;
    int tiles_x, tiles_per_img, wstr;
};

// RPW = output rows per wave (tile of 4 * RPW rows x 16 columns): 2 in general; 4 for cout = 16, whose waves otherwise do 36 MFMAs
// per (tile, channel chunk) item around two barriers and an LDS commit -- those layers ran at 69 TFLOP/s against the 106 of the
// 64-channel one (now 78; 128->16: 89 -> 105).  The k order per output pixel (chunk, tap, channel) does not depend on the tile: the
// results are bit-identical for every RPW.
template <int KS, int S, int MT, int RPW>
__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(ConvF32Args a) {
    constexpr int TH = 4 * RPW, TW = 16, CK = 8, TAPS = KS * KS, PAD = KS / 2;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS, NPIX = IH * IW;
    constexpr int NPAD = ((NPIX + 15) / 32) * 32 + 16;            // >= NPIX, = 16 mod 32
    static_assert(NPAD >= NPIX, "plane stride");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_x = (float*)smem;                      // [CK][NPAD]
    float* s_w = s_x + CK * NPAD;                   // [TAPS][CK][wstr]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int cin = a.c0 + a.c1, nchunk = cin / CK, wstr = a.wstr;
    const int batch = a.n_active ? *a.n_active : a.batch;
    const int total = a.tiles_per_img * (batch < a.batch ? batch : a.batch);
    constexpr int CQ = MT * 4;                                     // float4 units per weight row
    constexpr int W_UNITS = TAPS * CK * CQ, W_PT = (W_UNITS + 255) / 256;
    constexpr int X_UNITS = NPIX * 2, X_PT = (X_UNITS + 255) / 256;
    const int my_tiles = total > (int)blockIdx.x ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int n_items = my_tiles * nchunk;
    // Software pipeline over (tile, channel chunk) items: the global loads of item i+1 (input chunk + weight chunk) are issued
    // into registers before the MFMA loop of item i and written to LDS after it -- one memory round trip per item, hidden
    // behind the matrix work (a plain load / store loop costs one round trip per 256-unit slice: up to 11 per item).
    f32x4 px[X_PT], pw[W_PT];
    auto issue = [&](int item) {
        const int tl = blockIdx.x + (item / nchunk) * gridDim.x, chunk = item % nchunk;
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int gy0 = (t / a.tiles_x) * TH * S - PAD, gx0 = (t % a.tiles_x) * TW * S - PAD;
        const int cb = chunk * CK;
        const bool first = cb < a.c0;
        const float* src = first ? a.src0 : a.src1;
        const int csrc = first ? a.c0 : a.c1, ch0 = first ? cb : cb - a.c0;
#pragma unroll
        for (int k = 0; k < X_PT; ++k) {
            const int u = tid + k * 256;
            const int p = u >> 1, half = u & 1;
            const int gy = gy0 + p / IW, gx = gx0 + p % IW;
            px[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < X_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) px[k] = *(const f32x4*)(src + ((size_t)(b * a.H + gy) * a.W + gx) * csrc + ch0 + half * 4);
        }
#pragma unroll
        for (int k = 0; k < W_PT; ++k) {
            const int u = tid + k * 256;
            const int row = u / CQ, q = u % CQ;                   // row = tap * CK + c
            pw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < W_UNITS) pw[k] = *(const f32x4*)(a.w + ((size_t)((row / CK) * cin + cb + row % CK) * a.cout) + q * 4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < X_PT; ++k) {
            const int u = tid + k * 256;
            if (u < X_UNITS) {
                const int p = u >> 1, half = u & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) s_x[(half * 4 + j) * NPAD + p] = px[k][j];
            }
        }
#pragma unroll
        for (int k = 0; k < W_PT; ++k) {
            const int u = tid + k * 256;
            if (u < W_UNITS) *(f32x4*)(s_w + (u / CQ) * wstr + (u % CQ) * 4) = pw[k];
        }
    };
    f32x4 acc[MT][RPW];
    if (n_items > 0) issue(0);
    for (int item = 0; item < n_items; ++item) {
        const int tl = blockIdx.x + (item / nchunk) * gridDim.x, chunk = item % nchunk;
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;
        __syncthreads();                       // every wave finished reading the previous item's LDS image
        commit();
        __syncthreads();
        if (item + 1 < n_items) issue(item + 1);
        if (chunk == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[m][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int kk = 0; kk < CK / 4; ++kk) {
                const float* xp = s_x + (kk * 4 + g) * NPAD + ((RPW * wave) * S + dy) * IW + n * S + dx;
                float bq[RPW];
#pragma unroll
                for (int r = 0; r < RPW; ++r) bq[r] = xp[r * S * IW];
                const float* wp = s_w + (tap * CK + kk * 4 + g) * wstr + n;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float af = wp[m * 16];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bq[r], acc[m][r], 0, 0, 0);
                }
            }
        }
        if (chunk != nchunk - 1) continue;
        // epilogue: lane (n, g) holds couts m*16 + 4g .. +3 of pixel (row RPW*wave + t, column n)
#pragma unroll
        for (int t2 = 0; t2 < RPW; ++t2) {
            const int oy = oy0 + RPW * wave + t2, ox = ox0 + n;
            if (oy >= a.OH || ox >= a.OW) continue;
            const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.cout;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co = m * 16 + 4 * g;
                f32x4 v = acc[m][t2] + *(const f32x4*)(a.bias + co);
                if (a.residual) v += *(const f32x4*)(a.residual + o + co);
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                *(f32x4*)(a.dst + o + co) = v;
            }
        }
    }
}

template <int KS, int S, int MT, int RPW>
static int launch_f32_rpw(ConvF32Args& a, hipStream_t st);

template <int KS, int S, int MT>
static int launch_f32_t(ConvF32Args& a, hipStream_t st) {
    static const bool rpw2 = getenv("TTUP_F32_RPW2") != nullptr;          // cross-check: the 8-row tiles for every layer
    constexpr int RPW = MT == 1 ? 4 : 2;          // measured: cout 16 +12...17 % with 16-row tiles, cout 32 -5...-11 %
    if (RPW == 4 && rpw2) return launch_f32_rpw<KS, S, MT, 2>(a, st);
    // fewer 8-row tiles than CUs (the 1/8-resolution layers of one frame: 110 tiles): 4-row tiles fill twice as many CUs
    if (RPW == 2 && !rpw2 && (long long)cdiv(a.OW, 16) * cdiv(a.OH, 8) * a.batch < 256) return launch_f32_rpw<KS, S, MT, 1>(a, st);
    return launch_f32_rpw<KS, S, MT, RPW>(a, st);
}

template <int KS, int S, int MT, int RPW>
static int launch_f32_rpw(ConvF32Args& a, hipStream_t st) {
    constexpr int TH = 4 * RPW, TW = 16, CK = 8, TAPS = KS * KS;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS, NPIX = IH * IW;
    constexpr int NPAD = ((NPIX + 15) / 32) * 32 + 16;
    a.tiles_x = cdiv(a.OW, TW);
    a.tiles_per_img = a.tiles_x * cdiv(a.OH, TH);
    a.wstr = a.cout + (a.cout > 16 ? 16 : 0);
    const size_t smem = (size_t)(CK * NPAD + TAPS * CK * a.wstr) * sizeof(float);
    if (int rc = ensure_max_lds((const void*)conv_f32_mfma_kernel<KS, S, MT, RPW>, smem)) return rc;
    const long long total = (long long)a.tiles_per_img * a.batch;
    if (total == 0) return TTUP_OK;
    int per_cu = (int)((160 * 1024) / smem);
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    const int grid = total < 256 * per_cu ? (int)total : 256 * per_cu;
    hipLaunchKernelGGL((conv_f32_mfma_kernel<KS, S, MT, RPW>), dim3(grid), dim3(256), smem, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

template <int KS, int S>
static int dispatch_f32_mt(ConvF32Args& a, hipStream_t st) {
    switch (a.cout / 16) {
        case 1: return launch_f32_t<KS, S, 1>(a, st);
        case 2: return launch_f32_t<KS, S, 2>(a, st);
        case 4: return launch_f32_t<KS, S, 4>(a, st);
        case 8: return launch_f32_t<KS, S, 8>(a, st);
    }
    set_error("conv f32: cout %d unsupported", a.cout);
    return TTUP_EINVAL;
}

// true when the MFMA fp32 kernel handles this conv (everything in the HRNet graph; the direct kernel stays for the rest)
bool conv_f32_mfma_supported(const PackedConv& p) {
    const int mt = p.cout / 16;
    return p.cout % 16 == 0 && (mt == 1 || mt == 2 || mt == 4 || mt == 8) && p.cin_total % 8 == 0 && p.c0 % 8 == 0 &&
           (p.k == 1 || p.k == 3) && (p.stride == 1 || (p.stride == 2 && p.k == 3));
}

int launch_conv_f32_mfma(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    ConvF32Args a;
    a.src0 = (const float*)l.src0; a.src1 = (const float*)l.src1; a.w = (const float*)p.w_dev; a.bias = p.bias_dev;
    a.residual = (const float*)l.residual; a.dst = (float*)l.dst;
    a.c0 = p.c0; a.c1 = p.cin_total - p.c0; a.cout = p.cout; a.H = l.h; a.W = l.w;
    a.OH = (l.h + p.stride - 1) / p.stride; a.OW = (l.w + p.stride - 1) / p.stride; a.relu = l.relu; a.batch = l.batch;
    a.n_active = l.n_active;
    if (p.k == 3 && p.stride == 1) return dispatch_f32_mt<3, 1>(a, st);
    if (p.k == 3 && p.stride == 2) return dispatch_f32_mt<3, 2>(a, st);
    if (p.k == 1 && p.stride == 1) return dispatch_f32_mt<1, 1>(a, st);
    set_error("conv f32: k=%d stride=%d unsupported", p.k, p.stride);
    return TTUP_EINVAL;
}

}  // namespace ttup
