// a6/a7: the 2D->3D uplift transformer (reference uplifting/model.py 'connectstage', mode 'dynamic',
// time_rotation 'new') and the spin frame change (uplifting/helper.py:394-420), fp32 throughout.
//
// Structure (all tokens of a chunk of trajectories are processed as flat [tokens][D] arrays):
//   embed        BallEmbedding / TableEmbedding  model.py:105-158
//   table stage  (B*T) sequences x 14 tokens, 4 layers, RoPE on tokens 1..13 at fake times n/100 s  :360-384
//   time stage   B sequences x T tokens, depth-4 layers                                               :386-387
//   heads        MyHead 128->64->32->3                                                                :232-261
//   spin stage   cls token + T tokens, 4 layers, rotation head on the cls token                      :551-571
// Kernels:
//   linear_kernel   out = [relu](LN?(x) W^T + b) [+ res] on v_mfma_f32_16x16x4_f32 (exact fp32 products,
//                   fp32 accumulate).  W is the A operand (pre-packed per lane on the host), the token
//                   tile is the B operand read from LDS, so a lane owns 4 consecutive outputs of one token.
//   attention_kernel  per (sequence, head): RoPE(q,k) on load from a per-forward (cos,sin) table, additive {0,-inf}
//                   row+column mask, online softmax in registers; a fully masked query row yields zeros (torch SDPA
//                   semantics).  Short sequences (the 14-token table stage) share a wave four at a time.
#include "no_packed_fp32_begin.h"      // this unit's kernels run beside the CNN's chain kernels: no packed fp32 (common.h)
#include "common.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <map>
#include <type_traits>
#include <memory>
#include <utility>
#include <vector>

using namespace ttup;


namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------ packed linear layer
struct Linear {
    int n = 0, k = 0;            // out features, in features
    float* w_dev = nullptr;      // MFMA path: [ntile][k/16][64 lanes][4]; small-K path: [n][k] row major
    float* b_dev = nullptr;      // [n] or null
    bool mfma = false;
    // K = 128 layers (all of the 'large' model's transformer layers): the weights split into three bf16 parts, packed per
    // v_mfma_f32_16x16x32_bf16 A fragment: [ntile][k/32][plane][64 lanes][8] (linear_x3_kernel)
    uint16_t* w3_dev = nullptr;
};

// sum over the 16 lanes of a DPP row (every lane gets it): rotations by 8, 4, 2, 1 -- the same pairings, hence bit for bit the same
// value, as the xor butterfly of __shfl_xor, without its four trips through the LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight (a __syncthreads() drains vmcnt too)
__device__ __forceinline__ void stage_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct LinArgs {
    const float* x; int ldx;
    const float* w; const float* bias;
    const float* gamma; const float* beta;      // LayerNorm (null = none)
    const float* res; int ldr;
    float* out; int ldo;
    int M, N, K, relu;
};

// K permutation shared by the packed weights and the LDS image: MFMA k-step s, k-lane q  <->  k = q*(K/4) + s
// Workgroup tile: 64*MH token rows x 64*NTW outputs, 4*MH waves; wave (wm, wn) owns rows wm*64.. and N-tiles wn + 4t.
template <bool LN, int NTW, int MH>
__global__ __launch_bounds__(256 * MH) void linear_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [4][64*MH][K/4 + 4]
    constexpr int BM = 64 * MH;
    const int K = a.K, KQ = K / 4, RS = KQ + 4, PLANE = BM * RS;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
    const int m0 = ttup_bid_x() * BM, n0 = ttup_bid_y() * (64 * NTW);
    // ---- stage the token rows (LayerNorm applied on the way in): 16 lanes per row, float4 per lane per 64 features
    {
        const int grp = tid >> 4, l16 = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = grp + i * 16 * MH, m = m0 + r;
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * (l16 + 16 * u);
                v[u] = (m < a.M && k < K) ? *(const f32x4*)(a.x + (size_t)m * a.ldx + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (LN) {
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) sum += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
                sum = row16_sum(sum);
                const float mean = sum / (float)K;
                float var = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (4 * (l16 + 16 * u) >= K) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
                }
                var = row16_sum(var);
                const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 4 * (l16 + 16 * u);
                    if (k >= K) continue;
                    const f32x4 g = *(const f32x4*)(a.gamma + k), bt = *(const f32x4*)(a.beta + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[e] + bt[e];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * (l16 + 16 * u);
                if (k < K) *(f32x4*)(xs + (k / KQ) * PLANE + r * RS + (k % KQ)) = v[u];
            }
        }
    }
    __syncthreads();
    const int q = lane >> 4, c = lane & 15;
    f32x4 acc[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = (a.N + 15) / 16;
    int nt_g[NTW]; bool nt_ok[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { nt_g[t] = n0 / 16 + wn + 4 * t; nt_ok[t] = nt_g[t] < ntiles; }
    const int ks4 = K / 16;
    const float* xw = xs + q * PLANE + (wm * 64 + c) * RS;
    f32x4 wa[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
        wa[t] = nt_ok[t] ? *(const f32x4*)(a.w + (((size_t)nt_g[t] * ks4) * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s4 = 0; s4 < ks4; ++s4) {
        f32x4 xb[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xb[mt] = *(const f32x4*)(xw + mt * 16 * RS + s4 * 4);
        f32x4 wn_[NTW];
        const int sn = s4 + 1 < ks4 ? s4 + 1 : s4;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            wn_[t] = nt_ok[t] ? *(const f32x4*)(a.w + (((size_t)nt_g[t] * ks4 + sn) * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][j], xb[mt][j], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) wa[t] = wn_[t];
    }
    // ---- epilogue: lane holds outputs n = nt*16 + 4*q + {0..3} of token m = m0 + wm*64 + mt*16 + c
    const bool vec = (a.N % 4 == 0) && (a.ldo % 4 == 0) && (!a.res || a.ldr % 4 == 0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (!nt_ok[t]) continue;
        const int n = nt_g[t] * 16 + 4 * q;
        if (vec) {
            if (n >= a.N) continue;
            const f32x4 b4 = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + wm * 64 + mt * 16 + c;
                if (m >= a.M) continue;
                f32x4 v = acc[t][mt] + b4;
                if (a.relu) v = f32x4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
                if (a.res) v += *(const f32x4*)(a.res + (size_t)m * a.ldr + n);
                *(f32x4*)(a.out + (size_t)m * a.ldo + n) = v;
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wm * 64 + mt * 16 + c;
            if (m >= a.M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= a.N) continue;
                float v = acc[t][mt][r] + (a.bias ? a.bias[n + r] : 0.f);
                if (a.relu) v = v > 0.f ? v : 0.f;
                if (a.res) v += a.res[(size_t)m * a.ldr + n + r];
                a.out[(size_t)m * a.ldo + n + r] = v;
            }
        }
    }
}

// The same layer on the bf16 matrix pipe with SPLIT operands (the arithmetic of csrc/conv_x3.hip): every fp32 weight and every
// (LayerNorm'd) activation is split exactly into three bf16 parts, a product is the sum of six exact partial products (smallest
// first) accumulated in fp32 -- accurate to below one fp32 fma rounding, at 2.7x the peak rate of v_mfma_f32_16x16x4_f32.  K = 128
// only (the transformer layers of the 'large' model: 98 % of the work); TTUP_F32_EXACT=1 keeps the fp32-MFMA kernel.
// LDS image: three planes [64*MH tokens][128] bf16 (256-byte rows), the 16-byte chunk index XOR-swizzled with the token's low four
// bits: the 16 lanes of a ds_read_b128 group (8 tokens of one k chunk, 8 of the next) fall on 16 different chunks.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ unsigned ux3_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

template <bool LN, int NTW, int MH>
__global__ __launch_bounds__(256 * MH) void linear_x3_kernel(LinArgs a, const uint16_t* __restrict__ w3) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];      // [3][64*MH][128]
    constexpr int BM = 64 * MH, K = 128, PLANE = BM * K;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
    const int m0 = ttup_bid_x() * BM, n0 = ttup_bid_y() * (64 * NTW);
    // ---- stage the token rows (LayerNorm applied on the way in): 16 lanes per row, two float4 per lane (8 consecutive features)
    {
        const int grp = tid >> 4, l16 = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = grp + i * 16 * MH, m = m0 + r;
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) v[u] = m < a.M ? *(const f32x4*)(a.x + (size_t)m * a.ldx + 8 * l16 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (LN) {
                // (mean / variance with the summation tree of linear_kernel's staging is not required: any order is within the bar;
                // a 16-lane tree over 8 features per lane)
                float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
                sum = row16_sum(sum);
                const float mean = sum / (float)K;
                float var = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
                var = row16_sum(var);
                const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 g = *(const f32x4*)(a.gamma + 8 * l16 + 4 * u), bt = *(const f32x4*)(a.beta + 8 * l16 + 4 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[e] + bt[e];
                }
            }
            u32x4 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[j >> 1][2 * (j & 1)], x1 = v[j >> 1][2 * (j & 1) + 1];
                const unsigned q0 = ux3_pack2(x0, x1);
                const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
                const unsigned q1 = ux3_pack2(r0, r1);
                const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
                p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
            }
            uint16_t* d = xh + r * K + ((l16 ^ (r & 15)) << 3);
            *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
        }
    }
    __syncthreads();
    const int q = lane >> 4, c = lane & 15;
    f32x4 acc[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = (a.N + 15) / 16;
    int nt_g[NTW]; bool nt_ok[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { nt_g[t] = n0 / 16 + wn + 4 * t; nt_ok[t] = nt_g[t] < ntiles; }
    constexpr int KS = K / 32;
    // token c of m-tile mt sits in row wm*64 + mt*16 + c: (row & 15) == c, so the swizzle term is the lane's own c
    const uint16_t* xw = xh + (wm * 64 + c) * K;
    bf16x8 wa[3][NTW];
    const bf16x8 zero8 = {};
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) wa[p][t] = nt_ok[t] ? *(const bf16x8*)(w3 + ((((size_t)nt_g[t] * KS) * 3 + p) * 64 + lane) * 8) : zero8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bf16x8 xb[3][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[p][mt] = *(const bf16x8*)(xw + p * PLANE + mt * 16 * K + (((4 * s + q) ^ c) << 3));
        bf16x8 wn_[3][NTW];
        const int sn = s + 1 < KS ? s + 1 : s;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wn_[p][t] = nt_ok[t] ? *(const bf16x8*)(w3 + ((((size_t)nt_g[t] * KS + sn) * 3 + p) * 64 + lane) * 8) : zero8;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[PA[j]][t], xb[PB[j]][mt], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p][t] = wn_[p][t];
    }
    // ---- epilogue: lane holds outputs n = nt*16 + 4*q + {0..3} of token m = m0 + wm*64 + mt*16 + c  (N % 4 == 0 for K = 128 layers)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (!nt_ok[t]) continue;
        const int n = nt_g[t] * 16 + 4 * q;
        if (n >= a.N) continue;
        const f32x4 b4 = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wm * 64 + mt * 16 + c;
            if (m >= a.M) continue;
            f32x4 v = acc[t][mt] + b4;
            if (a.relu) v = f32x4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
            if (a.res) v += *(const f32x4*)(a.res + (size_t)m * a.ldr + n);
            *(f32x4*)(a.out + (size_t)m * a.ldo + n) = v;
        }
    }
}

// The token-local half of SimpleStaticLayer.forward (model.py:295-298) in ONE kernel, D = 128:
//     x2 = proj(att) + x;   hid = relu(fc1(LN(x2)));   x = fc2(hid) + x2
// Three chained 128 x 128 GEMMs (split-bf16 operands as in linear_x3_kernel) on a tile of 64*MH tokens; x2 stays in the registers of
// the lanes that produced it (the three GEMMs share one tiling, so the residual of the last one is already in place), LN(x2) and hid
// go through LDS, nothing but `att` and `x` is read and nothing but `x` written: 1.5 KB of HBM traffic per token instead of the
// 4.1 KB of the three separate launches (proj -> x2, fc1 -> hid, fc2 -> x), which at B = 10 000 trajectories are HBM-bound.
struct MlpArgs {
    const float* att; float* x; long long M;
    const uint16_t* w_proj; const uint16_t* w_fc1; const uint16_t* w_fc2;
    const float* g2; const float* b2; const float* bias1; const float* bias2;
};
template <int MH>
__global__ __launch_bounds__(256 * MH) void mlp_block_x3_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];      // [3][BM][128] split planes, then float s2[BM][132]
    constexpr int BM = 64 * MH, K = 128, PLANE = BM * K, KS = K / 32, NTW = 2;
    float* s2 = (float*)(xh + 3 * PLANE);                              // [BM][128] fp32, 16-byte chunks XOR-swizzled with the row's low 4 bits
    // (512-byte rows alias on the banks: the swizzle spreads the 8 rows of a ds_write_b128 lane group over 8 chunks; 80 KB per
    // 64-token workgroup = two per CU, 160 KB per 128-token workgroup)
    auto s2p = [&](int r, int n) __attribute__((always_inline)) { return s2 + r * K + ((((n >> 2) ^ (r & 15))) << 2); };
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
    const long long m0 = (long long)ttup_bid_x() * BM;
    const int q = lane >> 4, c = lane & 15;
    const int grp = tid >> 4, l16 = tid & 15;
    auto split_store = [&](int r, int chunk, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
        u32x4 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = j < 2 ? lo[2 * j] : hi[2 * (j - 2)], x1 = j < 2 ? lo[2 * j + 1] : hi[2 * (j - 2) + 1];
            const unsigned q0 = ux3_pack2(x0, x1);
            const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
            const unsigned q1 = ux3_pack2(r0, r1);
            const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
            p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
        }
        uint16_t* d = xh + r * K + ((chunk ^ (r & 15)) << 3);
        *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
    };
    auto gemm = [&](const uint16_t* __restrict__ w3, f32x4 (&acc)[NTW][4]) __attribute__((always_inline)) {
        const uint16_t* xw = xh + (wm * 64 + c) * K;
        bf16x8 wa[3][NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p][t] = *(const bf16x8*)(w3 + ((((size_t)(wn + 4 * t) * KS) * 3 + p) * 64 + lane) * 8);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 xb[3][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p][mt] = *(const bf16x8*)(xw + p * PLANE + mt * 16 * K + (((4 * s + q) ^ c) << 3));
            bf16x8 wn_[3][NTW];
            const int sn = s + 1 < KS ? s + 1 : s;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) wn_[p][t] = *(const bf16x8*)(w3 + ((((size_t)(wn + 4 * t) * KS + sn) * 3 + p) * 64 + lane) * 8);
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[PA[j]][t], xb[PB[j]][mt], acc[t][mt], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) wa[p][t] = wn_[p][t];
        }
    };
    // ---- 1. att rows -> split planes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = grp + i * 16 * MH;
        const long long m = m0 + r;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 lo = m < a.M ? *(const f32x4*)(a.att + m * K + 8 * l16) : z, hi = m < a.M ? *(const f32x4*)(a.att + m * K + 8 * l16 + 4) : z;
        split_store(r, l16, lo, hi);
    }
    __syncthreads();
    // ---- 2. x2 = proj(att) + x   (kept in registers; a copy goes to LDS for the LayerNorm)
    f32x4 x2[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) x2[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm(a.w_proj, x2);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (wn + 4 * t) * 16 + 4 * q;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int r = wm * 64 + mt * 16 + c;
            const long long m = m0 + r;
            if (m < a.M) x2[t][mt] += *(const f32x4*)(a.x + m * K + n);
            *(f32x4*)s2p(r, n) = x2[t][mt];
        }
    }
    __syncthreads();              // every wave is done reading the att planes; x2 rows are complete in s2
    // ---- 3. LN(x2) -> split planes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = grp + i * 16 * MH;
        f32x4 v[2] = {*(const f32x4*)s2p(r, 8 * l16), *(const f32x4*)s2p(r, 8 * l16 + 4)};
        float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
        sum = row16_sum(sum);
        const float mean = sum / (float)K;
        float var = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
        var = row16_sum(var);
        const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 g = *(const f32x4*)(a.g2 + 8 * l16 + 4 * u), bt = *(const f32x4*)(a.b2 + 8 * l16 + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[e] + bt[e];
        }
        split_store(r, l16, v[0], v[1]);
    }
    __syncthreads();
    // ---- 4. hid = relu(fc1(LN(x2)) + b1) -> split planes (through s2: a lane holds 4 outputs of a token, a chunk is 8)
    {
        f32x4 acc[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(a.w_fc1, acc);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int n = (wn + 4 * t) * 16 + 4 * q;
            const f32x4 b4 = *(const f32x4*)(a.bias1 + n);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 v = acc[t][mt] + b4;
                v = f32x4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
                *(f32x4*)s2p(wm * 64 + mt * 16 + c, n) = v;          // (s2's LayerNorm input has been consumed: barrier above)
            }
        }
    }
    __syncthreads();              // GEMM 2 has read its planes; hid rows are complete in s2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = grp + i * 16 * MH;
        split_store(r, l16, *(const f32x4*)s2p(r, 8 * l16), *(const f32x4*)s2p(r, 8 * l16 + 4));
    }
    __syncthreads();
    // ---- 5. x = fc2(hid) + b2 + x2
    {
        f32x4 acc[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(a.w_fc2, acc);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int n = (wn + 4 * t) * 16 + 4 * q;
            const f32x4 b4 = *(const f32x4*)(a.bias2 + n);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const long long m = m0 + wm * 64 + mt * 16 + c;
                if (m < a.M) *(f32x4*)(a.x + m * K + n) = (acc[t][mt] + b4) + x2[t][mt];
            }
        }
    }
}

// The same block in the form of stage_x3_kernel's MLP half (round 4, after that kernel turned out twice as fast per tile): 8 waves on
// a 64-token tile, wave w owns output features 16 w .. 16 w + 15 of all 64 rows in each of the three GEMMs (no weight fragment is
// fetched twice by a workgroup), its 12 KB weight tile of the NEXT GEMM is requested before the current one starts and stays in
// flight across the LDS phases (LDS-only barriers, loads pinned with scheduling barriers), LayerNorm row sums on DPP.  80 KB of LDS:
// two workgroups per CU, whose phases interleave.  Arithmetic identical to mlp_block_x3_kernel (same split, same order per output).
__global__ __launch_bounds__(512) void mlp_block8_x3_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];      // [3][64][128] split planes | float s2[64][128] (swizzled)
    constexpr int BM = 64, K = 128, PLANE = BM * K, KS = K / 32;
    float* s2 = (float*)(xh + 3 * PLANE);
    auto swz = [&](float* b, int r, int n) __attribute__((always_inline)) { return b + r * K + ((((n >> 2) ^ (r & 15))) << 2) + (n & 3); };
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)ttup_bid_x() * BM;
    const int q = lane >> 4, c = lane & 15;
    const int grp = tid >> 4, l16 = tid & 15;
    const int n = wave * 16 + 4 * q;
    auto split_store = [&](int r, int chunk, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
        u32x4 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = j < 2 ? lo[2 * j] : hi[2 * (j - 2)], x1 = j < 2 ? lo[2 * j + 1] : hi[2 * (j - 2) + 1];
            const unsigned q0 = ux3_pack2(x0, x1);
            const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
            const unsigned q1 = ux3_pack2(r0, r1);
            const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
            p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
        }
        uint16_t* d = xh + r * K + ((chunk ^ (r & 15)) << 3);
        *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
    };
    auto load_tile = [&](const uint16_t* __restrict__ w3, bf16x8 (&w)[3][KS]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p][s] = *(const bf16x8*)(w3 + ((((size_t)wave * KS + s) * 3 + p) * 64 + lane) * 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gemm = [&](const bf16x8 (&w)[3][KS], f32x4 (&acc)[4]) __attribute__((always_inline)) {
        const uint16_t* xw = xh + c * K;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 xb[3][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p][mt] = *(const bf16x8*)(xw + p * PLANE + mt * 16 * K + (((4 * s + q) ^ c) << 3));
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[PA[j]][s], xb[PB[j]][mt], acc[mt], 0, 0, 0);
        }
    };
    // ---- small operands first (the memory counter retires in order), then the first weight tile, then the att rows
    f32x4 xr[4], lg[2], lb[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const long long m = m0 + mt * 16 + c;
        xr[mt] = m < a.M ? *(const f32x4*)(a.x + m * K + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { lg[u] = *(const f32x4*)(a.g2 + 8 * l16 + 4 * u); lb[u] = *(const f32x4*)(a.b2 + 8 * l16 + 4 * u); }
    const f32x4 bias1 = *(const f32x4*)(a.bias1 + n), bias2 = *(const f32x4*)(a.bias2 + n);
    f32x4 at[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + grp + 32 * i;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        at[i][0] = m < a.M ? *(const f32x4*)(a.att + m * K + 8 * l16) : z;
        at[i][1] = m < a.M ? *(const f32x4*)(a.att + m * K + 8 * l16 + 4) : z;
    }
    bf16x8 wnext[3][KS];
    load_tile(a.w_proj, wnext);
#pragma unroll
    for (int i = 0; i < 2; ++i) split_store(grp + 32 * i, l16, at[i][0], at[i][1]);
    stage_barrier();
    // ---- x2 = proj(att) + x
    f32x4 x2[4];
    {
        bf16x8 wc[3][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
        load_tile(a.w_fc1, wnext);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) x2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(wc, x2);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            x2[mt] += xr[mt];
            *(f32x4*)swz(s2, mt * 16 + c, n) = x2[mt];
        }
    }
    stage_barrier();
    // ---- LN(x2) -> split planes
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = grp + 32 * i;
        f32x4 v[2] = {*(const f32x4*)swz(s2, r, 8 * l16), *(const f32x4*)swz(s2, r, 8 * l16 + 4)};
        float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
        sum = row16_sum(sum);
        const float mean = sum / (float)K;
        float var = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
        var = row16_sum(var);
        const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * lg[u][e] + lb[u][e];
        split_store(r, l16, v[0], v[1]);
    }
    stage_barrier();
    // ---- hid = relu(fc1(LN(x2)) + b1) -> s2 -> split planes
    {
        bf16x8 wc[3][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
        load_tile(a.w_fc2, wnext);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(wc, acc);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 v = acc[mt] + bias1;
            v = f32x4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
            *(f32x4*)swz(s2, mt * 16 + c, n) = v;
        }
    }
    stage_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = grp + 32 * i;
        split_store(r, l16, *(const f32x4*)swz(s2, r, 8 * l16), *(const f32x4*)swz(s2, r, 8 * l16 + 4));
    }
    stage_barrier();
    // ---- x = fc2(hid) + b2 + x2
    {
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(wnext, acc);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const long long m = m0 + mt * 16 + c;
            if (m < a.M) *(f32x4*)(a.x + m * K + n) = (acc[mt] + bias2) + x2[mt];
        }
    }
}

// qkv = LN(x) Wqkv^T + b for D = 128 (384 outputs), the stage kernel's steps 1-2 with the result written to memory: 8 waves on a
// 64-token tile, wave w computes the q, k and v tiles w, 8 + w, 16 + w (all 64 rows each), weight tiles requested one GEMM ahead.
// Used instead of linear_x3_kernel<true, 3, *> for launches of at most 256 tiles (the hub surface, the pipeline's per-clip uplift),
// where one workgroup's latency is what counts.
struct QkvArgs { const float* x; float* qkv; long long M; const uint16_t* w_qkv; const float* b_qkv; const float* g1; const float* b1; };
__global__ __launch_bounds__(512) void qkv_block8_x3_kernel(QkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];      // [3][64][128] split planes
    constexpr int BM = 64, K = 128, PLANE = BM * K, KS = K / 32;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)ttup_bid_x() * BM;
    const int q = lane >> 4, c = lane & 15;
    const int grp = tid >> 4, l16 = tid & 15;
    auto load_tile = [&](int nt, bf16x8 (&w)[3][KS]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p][s] = *(const bf16x8*)(a.w_qkv + ((((size_t)nt * KS + s) * 3 + p) * 64 + lane) * 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- LN(x) rows -> split planes (16 lanes per row, rows grp and grp + 32)
    f32x4 xv[2][2], lg[2], lb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + grp + 32 * i;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        xv[i][0] = m < a.M ? *(const f32x4*)(a.x + m * K + 8 * l16) : z;
        xv[i][1] = m < a.M ? *(const f32x4*)(a.x + m * K + 8 * l16 + 4) : z;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { lg[u] = *(const f32x4*)(a.g1 + 8 * l16 + 4 * u); lb[u] = *(const f32x4*)(a.b1 + 8 * l16 + 4 * u); }
    bf16x8 wnext[3][KS];
    load_tile(wave, wnext);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = grp + 32 * i;
        f32x4 (&v)[2] = xv[i];
        float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
        sum = row16_sum(sum);
        const float mean = sum / (float)K;
        float var = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
        var = row16_sum(var);
        const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
        u32x4 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = (v[j >> 1][2 * (j & 1)] - mean) * rstd * lg[j >> 1][2 * (j & 1)] + lb[j >> 1][2 * (j & 1)];
            const float x1 = (v[j >> 1][2 * (j & 1) + 1] - mean) * rstd * lg[j >> 1][2 * (j & 1) + 1] + lb[j >> 1][2 * (j & 1) + 1];
            const unsigned q0 = ux3_pack2(x0, x1);
            const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
            const unsigned q1 = ux3_pack2(r0, r1);
            const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
            p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
        }
        uint16_t* d = xh + r * K + ((l16 ^ (r & 15)) << 3);
        *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
    }
    stage_barrier();
    // ---- the wave's q, k, v tiles
    const uint16_t* xw = xh + c * K;
#pragma unroll
    for (int jp = 0; jp < 3; ++jp) {
        bf16x8 wc[3][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
        const int nn = jp * K + wave * 16 + 4 * q;
        const f32x4 b4 = *(const f32x4*)(a.b_qkv + nn);
        if (jp < 2) load_tile((jp + 1) * 8 + wave, wnext);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 xb[3][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p][mt] = *(const bf16x8*)(xw + p * PLANE + mt * 16 * K + (((4 * s + q) ^ c) << 3));
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[PA[j]][s], xb[PB[j]][mt], acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const long long m = m0 + mt * 16 + c;
            if (m < a.M) *(f32x4*)(a.qkv + m * (3 * K) + nn) = acc[mt] + b4;
        }
    }
}

// out[m][n] = relu?(sum_k x[m][k] w[n][k] + b[n]) for tiny K (2 or 3): embedding fc1
__global__ void small_linear_kernel(const float* x, int ldx, const float* w, const float* b, float* out, int ldo, long long M, int N, int K, int relu) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= M * N) return;
    const long long m = i / N; const int n = (int)(i % N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(x[m * ldx + k], w[n * K + k], acc);
    acc += b ? b[n] : 0.f;
    if (relu) acc = acc > 0.f ? acc : 0.f;
    out[m * ldo + n] = acc;
}

// ------------------------------------------------------------------ attention
// rope[r][i] = (cos, sin)(round(t_r / 0.002) * inv_freq[i]) for every time stamp r           (model.py:62-80)
__global__ void rope_table_kernel(const float* times, const float* inv_freq, float2* rope, int half, long long total) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= total) return;
    const float pos = rintf(times[i / half] / 0.002f);          // round(t / (1/MAX_FPS)), model.py:72
    const float f = pos * inv_freq[i % half];
    rope[i] = make_float2(cosf(f), sinf(f));
}

struct AttnArgs {
    const float* qkv;   // [n_seq*S][3D]
    float* out;         // [n_seq*S][D]
    const float* mask;  // additive, row = seq / mask_div, S entries
    const float2* rope; // (cos, sin) rows of hd/2; row of token j = (seq / times_div) * times_stride + j - num_cls
    int n_seq, S, D, heads, hd, num_cls, mask_div, times_div, times_stride;
    float scale;
};

// P threads per (sequence, head); a workgroup of ttup_bdim_x() threads serves ttup_bdim_x() / P sequences.  K (rotated) and V
// of each sequence live in LDS, thread i0 owns query rows i0, i0+P, ...
template <int HD, int P>
__global__ __launch_bounds__(128) void attention_kernel(AttnArgs a) {      // at most 128 threads are ever launched: 256 VGPRs, no spills
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = a.S, G = ttup_bdim_x() / P;
    const int SEQ = 2 * S * HD + 16;                 // floats per sequence; the +16 words spreads the groups over LDS banks
    float* ms = sm + G * SEQ;                        // [G][S] additive mask
    const int h = ttup_bid_y(), tid = ttup_tid_x();
    const int D3 = 3 * a.D, HV = HD / 4;
    // ---- stage K (RoPE applied) and V, one float4 per thread per step, 128 B rows read by HV consecutive threads
    for (int u = tid; u < G * S * HV; u += ttup_bdim_x()) {
        const int g = u / (S * HV), rem = u - g * (S * HV), j = rem / HV, part = rem - j * HV;
        const int seq = ttup_bid_x() * G + g;
        if (seq >= a.n_seq) continue;
        const float* kp = a.qkv + ((size_t)seq * S + j) * D3 + a.D + h * HD + part * 4;
        f32x4 k = *(const f32x4*)kp;
        const f32x4 v = *(const f32x4*)(kp + a.D);
        if (j >= a.num_cls) {
            const f32x4 cs = *(const f32x4*)(a.rope + ((size_t)(seq / a.times_div) * a.times_stride + (j - a.num_cls)) * (HD / 2) + part * 2);
            k = f32x4{k[0] * cs[0] - k[1] * cs[1], k[0] * cs[1] + k[1] * cs[0], k[2] * cs[2] - k[3] * cs[3], k[2] * cs[3] + k[3] * cs[2]};
        }
        *(f32x4*)(sm + g * SEQ + j * HD + part * 4) = k;
        *(f32x4*)(sm + g * SEQ + S * HD + j * HD + part * 4) = v;
    }
    for (int u = tid; u < G * S; u += ttup_bdim_x()) {
        const int seq = ttup_bid_x() * G + u / S;
        ms[u] = seq < a.n_seq ? a.mask[(size_t)(seq / a.mask_div) * S + (u % S)] : -INFINITY;
    }
    __syncthreads();
    const int g = tid / P, i0 = tid - g * P;
    const int seq = ttup_bid_x() * G + g;
    if (seq >= a.n_seq) return;
    const float* ks = sm + g * SEQ;
    const float* vs = ks + S * HD;
    const float* mg = ms + g * S;
    for (int i = i0; i < S; i += P) {
        f32x4 q[HV];
        const float* qp = a.qkv + ((size_t)seq * S + i) * D3 + h * HD;
#pragma unroll
        for (int d = 0; d < HV; ++d) q[d] = *(const f32x4*)(qp + 4 * d);
        if (i >= a.num_cls) {
            const float2* rp = a.rope + ((size_t)(seq / a.times_div) * a.times_stride + (i - a.num_cls)) * (HD / 2);
#pragma unroll
            for (int d = 0; d < HV; ++d) {
                const f32x4 cs = *(const f32x4*)(rp + 2 * d);
                q[d] = f32x4{q[d][0] * cs[0] - q[d][1] * cs[1], q[d][0] * cs[1] + q[d][1] * cs[0],
                             q[d][2] * cs[2] - q[d][3] * cs[3], q[d][2] * cs[3] + q[d][3] * cs[2]};
            }
        }
        f32x4 o[HV];
#pragma unroll
        for (int d = 0; d < HV; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        float mx = -INFINITY, den = 0.f;
        if (mg[i] == 0.f) {
            for (int j = 0; j < S; ++j) {
                if (mg[j] != 0.f) continue;             // -inf column
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HV; ++d) {
                    const f32x4 kk = *(const f32x4*)(ks + j * HD + 4 * d);
                    s = fmaf(q[d][0], kk[0], s); s = fmaf(q[d][1], kk[1], s); s = fmaf(q[d][2], kk[2], s); s = fmaf(q[d][3], kk[3], s);
                }
                s *= a.scale;
                if (s > mx) {
                    const float corr = expf(mx - s);
                    den *= corr;
#pragma unroll
                    for (int d = 0; d < HV; ++d) o[d] *= corr;
                    mx = s;
                }
                const float p = expf(s - mx);
                den += p;
#pragma unroll
                for (int d = 0; d < HV; ++d) {
                    const f32x4 vv = *(const f32x4*)(vs + j * HD + 4 * d);
                    o[d][0] = fmaf(p, vv[0], o[d][0]); o[d][1] = fmaf(p, vv[1], o[d][1]);
                    o[d][2] = fmaf(p, vv[2], o[d][2]); o[d][3] = fmaf(p, vv[3], o[d][3]);
                }
            }
        }
        float* op = a.out + ((size_t)seq * S + i) * a.D + h * HD;
        const float inv = den > 0.f ? 1.f / den : 0.f;
#pragma unroll
        for (int d = 0; d < HV; ++d) *(f32x4*)(op + 4 * d) = o[d] * inv;
    }
}

// ------------------------------------------------------------------ attention on the fp32 matrix pipe, long sequences
// The temporal / spin stages (sequences of T or T+1 tokens, head dim 32).  attention_kernel walks the keys with one thread per query
// row -- 121 dependent exp / fma rounds: 60-70 us for a single rally, 15 % of the time at B = 10 000.  Here a wave owns 16 queries
// of one (sequence, head): K (RoPE applied) and V of the whole sequence are staged in LDS once per workgroup (4 waves = 64 queries);
// per 16-key tile  scores^T = K Q^T  (8 v_mfma_f32_16x16x4_f32: a lane ends with the scores of ONE query against four keys, so the
// row maximum and the denominator are in-lane sums plus two cross-lane steps at the end) in a first pass for the maxima, and again in
// a second pass for p = exp(s - max) and  out += P V  (8 more MFMAs, key index permuted so that p is already the A operand).  The
// normalisation 1 / den goes through 16 floats of LDS (out rows are indexed by 4q + r, den by the lane's own query).
struct AttnMArgs {
    const float* qkv; float* out; const float* mask; const float2* rope;
    int n_seq, S, num_cls, mask_div, times_div, times_stride;
    float scale;
};
constexpr int ATTM_KS = 36;          // floats per K / V row in LDS (144 B: 16 consecutive rows fall on 16 different 16-byte slots)
__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnMArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // K [SP][36] | V [SP][36] | inv [4 waves][16]
    constexpr int HD = 32, D = 128, D3 = 384, KS = ATTM_KS;
    const int S = a.S, KT = (S + 15) / 16, SP = KT * 16;
    float* sk = sm;
    float* sv = sm + SP * KS;
    float* sinv = sv + SP * KS;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int h = ttup_bid_y(), seq = ttup_bid_z();
    const float* base = a.qkv + (size_t)seq * S * D3 + h * HD;
    const float2* rbase = a.rope + (size_t)(seq / a.times_div) * a.times_stride * (HD / 2);
    const float* mrow = a.mask + (size_t)(seq / a.mask_div) * S;
    // ---- stage K (rotated) and V: 8 threads per row, one float4 each; rows past S are zero
    for (int u = tid; u < SP * 8; u += 256) {
        const int j = u >> 3, part = u & 7;
        f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
        if (j < S) {
            k = *(const f32x4*)(base + (size_t)j * D3 + D + part * 4);
            v = *(const f32x4*)(base + (size_t)j * D3 + 2 * D + part * 4);
            if (j >= a.num_cls) {
                const f32x4 cs = *(const f32x4*)(rbase + (size_t)(j - a.num_cls) * (HD / 2) + part * 2);
                k = f32x4{k[0] * cs[0] - k[1] * cs[1], k[0] * cs[1] + k[1] * cs[0], k[2] * cs[2] - k[3] * cs[3], k[2] * cs[3] + k[3] * cs[2]};
            }
        }
        *(f32x4*)(sk + j * KS + part * 4) = k;
        *(f32x4*)(sv + j * KS + part * 4) = v;
    }
    __syncthreads();
    const int qt = ttup_bid_x() * 4 + wave;                   // this wave's tile of 16 queries
    if (qt * 16 >= S) return;                                // (no barrier below: waves are independent from here on)
    const int i = qt * 16 + c;                               // the lane's query
    const bool row_ok = i < S && mrow[i < S ? i : 0] == 0.f;
    // B operand of scores^T: Q[i][8q .. 8q+7], rotated
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
    if (i < S) {
        q0 = *(const f32x4*)(base + (size_t)i * D3 + 8 * q);
        q1 = *(const f32x4*)(base + (size_t)i * D3 + 8 * q + 4);
        if (i >= a.num_cls) {
            const float2* rp = rbase + (size_t)(i - a.num_cls) * (HD / 2) + 4 * q;
            const f32x4 c0 = *(const f32x4*)rp, c1 = *(const f32x4*)(rp + 2);
            q0 = f32x4{q0[0] * c0[0] - q0[1] * c0[1], q0[0] * c0[1] + q0[1] * c0[0], q0[2] * c0[2] - q0[3] * c0[3], q0[2] * c0[3] + q0[3] * c0[2]};
            q1 = f32x4{q1[0] * c1[0] - q1[1] * c1[1], q1[0] * c1[1] + q1[1] * c1[0], q1[2] * c1[2] - q1[3] * c1[3], q1[2] * c1[3] + q1[3] * c1[2]};
        }
    }
    auto scores = [&](int kt) __attribute__((always_inline)) {
        // A operand: K[kt*16 + c][8q .. 8q+7]; result sc[r] = q_i . k_j for j = kt*16 + 4q + r, masked keys -> -inf
        const float* kp = sk + (kt * 16 + c) * KS + 8 * q;
        const f32x4 k0 = *(const f32x4*)kp, k1 = *(const f32x4*)(kp + 4);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[e], q0[e], sc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[e], q1[e], sc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = kt * 16 + 4 * q + r;
            const bool col_ok = j < S && mrow[j < S ? j : 0] == 0.f;
            sc[r] = col_ok ? sc[r] * a.scale : -INFINITY;
        }
        return sc;
    };
    float mx = -INFINITY;
    for (int kt = 0; kt < KT; ++kt) {
        const f32x4 sc = scores(kt);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = sc[r] > mx ? sc[r] : mx;
    }
    { const float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; }
    { const float o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int kt = 0; kt < KT; ++kt) {
        const f32x4 sc = scores(kt);
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = (row_ok && sc[r] > -INFINITY) ? __expf(sc[r] - mx) : 0.f; den += pr[r]; }
        // out += P V with k index (step s, lane group q) <-> key kt*16 + 4q + s: the A operand of step s is the lane's own pr[s]
        const float* vp = sv + (kt * 16 + 4 * q) * KS + c;
#pragma unroll
        for (int s2_ = 0; s2_ < 4; ++s2_) {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vp[s2_ * KS], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vp[s2_ * KS + 16], o1, 0, 0, 0);
        }
    }
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    // o[r] = out[query qt*16 + 4q + r][dim c (o0) / 16 + c (o1)]: the row's 1 / den comes from the lane that owns that query
    if (q == 0) sinv[wave * 16 + c] = den > 0.f ? 1.f / den : 0.f;          // a fully masked query row yields zeros (torch SDPA semantics)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes are visible to its reads
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int io = qt * 16 + 4 * q + r;
        if (io >= S) continue;
        const float inv = sinv[wave * 16 + 4 * q + r];
        float* op = a.out + ((size_t)seq * S + io) * D + h * HD;
        op[c] = o0[r] * inv;
        op[16 + c] = o1[r] * inv;
    }
}

// The same attention for sequences of at most 128 tokens (KT <= 8 key tiles: the 121-token trajectories of the headline and of config 3)
// in the form the stage kernel's attention phase arrived at: all score tiles of a query tile are computed ONCE, as independent MFMA
// chains (groups of four key tiles), and stay in registers between the maximum and the exponentials; the mask is two ballots per wave
// instead of a global load per score; V is staged TRANSPOSED ([dim][token], row stride SP + 4) so that the P V operand of four keys is
// one 16-byte read; exponentials on v_exp_f32.  Per output the operation order is attention_mfma_kernel's.  NG = groups of four key tiles.
template <int NG>
__global__ __launch_bounds__(256) void attention_mfma8_kernel(AttnMArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // K [SP][36] | V^T [32][SP + 4] | inv [4 waves][16]
    constexpr int HD = 32, D = 128, D3 = 384, KS = ATTM_KS, NK = NG * 4;
    const int S = a.S, KT = (S + 15) / 16, SP = KT * 16, VS = SP + 4;
    float* sk = sm;
    float* svt = sm + SP * KS;
    float* sinv = svt + HD * VS;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int h = ttup_bid_y(), seq = ttup_bid_z();
    const float* base = a.qkv + (size_t)seq * S * D3 + h * HD;
    const float2* rbase = a.rope + (size_t)(seq / a.times_div) * a.times_stride * (HD / 2);
    const float* mrow = a.mask + (size_t)(seq / a.mask_div) * S;
    // ---- stage K (rotated) and V^T: 8 threads per token, one float4 of each per thread; tokens past S are zero
    for (int u = tid; u < SP * 8; u += 256) {
        const int j = u >> 3, part = u & 7;
        f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
        if (j < S) {
            k = *(const f32x4*)(base + (size_t)j * D3 + D + part * 4);
            v = *(const f32x4*)(base + (size_t)j * D3 + 2 * D + part * 4);
            if (j >= a.num_cls) {
                const f32x4 cs = *(const f32x4*)(rbase + (size_t)(j - a.num_cls) * (HD / 2) + part * 2);
                k = f32x4{k[0] * cs[0] - k[1] * cs[1], k[0] * cs[1] + k[1] * cs[0], k[2] * cs[2] - k[3] * cs[3], k[2] * cs[3] + k[3] * cs[2]};
            }
        }
        *(f32x4*)(sk + j * KS + part * 4) = k;
#pragma unroll
        for (int e = 0; e < 4; ++e) svt[(part * 4 + e) * VS + j] = v[e];
    }
    // bit j of (lo, hi): token j / 64 + j is a valid key and query
    const unsigned long long lo = __builtin_amdgcn_ballot_w64(lane < S && mrow[lane < S ? lane : 0] == 0.f);
    const unsigned long long hi = __builtin_amdgcn_ballot_w64(64 + lane < S && mrow[64 + lane < S ? 64 + lane : 0] == 0.f);
    __syncthreads();
    const int qt = ttup_bid_x() * 4 + wave;                 // this wave's tile of 16 queries
    if (qt * 16 >= S) return;                                // (no barrier below: waves are independent from here on)
    const int i = qt * 16 + c;                               // the lane's query
    const bool row_ok = i < S && (((i < 64 ? lo : hi) >> (i & 63)) & 1);
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
    if (i < S) {
        q0 = *(const f32x4*)(base + (size_t)i * D3 + 8 * q);
        q1 = *(const f32x4*)(base + (size_t)i * D3 + 8 * q + 4);
        if (i >= a.num_cls) {
            const float2* rp = rbase + (size_t)(i - a.num_cls) * (HD / 2) + 4 * q;
            const f32x4 c0 = *(const f32x4*)rp, c1 = *(const f32x4*)(rp + 2);
            q0 = f32x4{q0[0] * c0[0] - q0[1] * c0[1], q0[0] * c0[1] + q0[1] * c0[0], q0[2] * c0[2] - q0[3] * c0[3], q0[2] * c0[3] + q0[3] * c0[2]};
            q1 = f32x4{q1[0] * c1[0] - q1[1] * c1[1], q1[0] * c1[1] + q1[1] * c1[0], q1[2] * c1[2] - q1[3] * c1[3], q1[2] * c1[3] + q1[3] * c1[2]};
        }
    }
    // ---- scores^T = K Q^T for every key tile (tiles past KT repeat the last one and are masked: their bits are 0)
    f32x4 sc[NK];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        f32x4 kk[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kt = g * 4 + t < KT ? g * 4 + t : KT - 1;
            const float* kp = sk + (kt * 16 + c) * KS + 8 * q;
            kk[t][0] = *(const f32x4*)kp; kk[t][1] = *(const f32x4*)(kp + 4);
            sc[g * 4 + t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[g * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[t][0][e], q0[e], sc[g * 4 + t], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[g * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[t][1][e], q1[e], sc[g * 4 + t], 0, 0, 0);
    }
    // sc[kt][r] = q_i . k_j for j = kt*16 + 4q + r
    const unsigned long long lo_q = lo >> (4 * q), hi_q = hi >> (4 * q);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool col_ok = kt < KT && (((kt < 4 ? lo_q : hi_q) >> ((kt & 3) * 16 + r)) & 1);
            sc[kt][r] = col_ok ? sc[kt][r] * a.scale : -INFINITY;
            mx = sc[kt][r] > mx ? sc[kt][r] : mx;
        }
    { const float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; }
    { const float o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
    // ---- p = exp(s - max), out += P V with k index (step s, lane group q) <-> key kt*16 + 4q + s: the A operand is the lane's own p
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    const float* vbase = svt + c * VS + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        f32x4 vv[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kt = g * 4 + t < KT ? g * 4 + t : KT - 1;
            vv[t][0] = *(const f32x4*)(vbase + kt * 16); vv[t][1] = *(const f32x4*)(vbase + 16 * VS + kt * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = (row_ok && sc[g * 4 + t][r] > -INFINITY) ? __expf(sc[g * 4 + t][r] - mx) : 0.f; den += pr[r]; }
#pragma unroll
            for (int s2_ = 0; s2_ < 4; ++s2_) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vv[t][0][s2_], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vv[t][1][s2_], o1, 0, 0, 0);
            }
        }
    }
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    if (q == 0) sinv[wave * 16 + c] = den > 0.f ? 1.f / den : 0.f;          // a fully masked query row yields zeros (torch SDPA semantics)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes are visible to its reads
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int io = qt * 16 + 4 * q + r;
        if (io >= S) continue;
        const float inv = sinv[wave * 16 + 4 * q + r];
        float* op = a.out + ((size_t)seq * S + io) * D + h * HD;
        op[c] = o0[r] * inv;
        op[16 + c] = o1[r] * inv;
    }
}

// ------------------------------------------------------------------ fused attention half of a layer, short sequences
// att = softmax-attention(RoPE(q), RoPE(k), v) with qkv = LN(x) Wqkv^T + b, for sequences of S <= 16 tokens (the table stage: 14),
// D = 128, 4 heads of 32: ONE kernel instead of the qkv linear + the attention launch, and the 1536 bytes of qkv per token never
// leave the CU (at B = 10 000 trajectories the table stage is 17 M tokens per layer).  A workgroup of 8 waves owns SEQS = 64 / S
// whole sequences (rows beyond SEQS*S idle):
//   1. LN(x) rows -> three split-bf16 planes in LDS; every wave loads the twelve fragments of ITS 16 rows into registers (the plane
//      storage is free after that and is reused for qkv);
//   2. qkv of all four heads by the split-bf16 GEMM of linear_x3_kernel: wave (m-tile w & 3, head pair w >> 2) streams the weight
//      fragments of its 12 n-tiles from L2; bias and RoPE (q, k; not the cls rows) in the epilogue -> LDS [64][4 x (q|k|v)];
//   3. attention on the fp32 matrix pipe, one (sequence, head) per wave at a time: scores^T = K Q^T (8 v_mfma_f32_16x16x4_f32: a lane
//      ends with the scores of ONE query against four keys, so the softmax is in-lane plus two cross-lane steps), P V with the key
//      index permuted so that the probabilities are already where the A operand wants them (8 more MFMAs) -> att.
// (First version: scalar attention, four threads per query row -- VALU-bound on redundant exp() calls, no faster than the two
// separate launches.)
struct AttnBlockArgs {
    const float* x; float* att; long long n_seq;
    const uint16_t* w_qkv; const float* b_qkv; const float* g1; const float* b1;
    const float* mask; const float2* rope;
    int S, num_cls, mask_div, times_div, times_stride;
    float scale;
};
constexpr int ATTN_QS = 196;          // floats per row of the qkv tile: 2 heads x 96 + 4 (784 B = 49 slots of 16 B: consecutive rows fall on consecutive slots)
__global__ __launch_bounds__(512) void attn_block_x3_kernel(AttnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];      // [3][64][128] split planes of LN(x); then float qh[64][ATTN_QS], two heads at a time
    // (50 KB: two workgroups per CU; with all four heads in LDS -- 99 KB, one workgroup per CU -- the kernel was latency-bound)
    constexpr int BM = 64, K = 128, PLANE = BM * K, KS = K / 32, HD = 32, QS = ATTN_QS;
    float* qh = (float*)xh;
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const int S = a.S, SEQS = BM / S, ROWS = SEQS * S;
    const long long seq0 = (long long)ttup_bid_x() * SEQS;
    const long long m0 = seq0 * S, M = a.n_seq * S;
    const int q = lane >> 4, c = lane & 15;
    // ---- 1. LN(x) rows -> split planes (16 lanes per row, 32 rows per pass)
    {
        const int grp = tid >> 4, l16 = tid & 15;
        const f32x4 gg[2] = {*(const f32x4*)(a.g1 + 8 * l16), *(const f32x4*)(a.g1 + 8 * l16 + 4)};
        const f32x4 bb[2] = {*(const f32x4*)(a.b1 + 8 * l16), *(const f32x4*)(a.b1 + 8 * l16 + 4)};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = grp + i * 32;
            const long long m = m0 + r;
            const bool ok = r < ROWS && m < M;
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) v[u] = ok ? *(const f32x4*)(a.x + m * K + 8 * l16 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
            float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
            sum = row16_sum(sum);
            const float mean = sum / (float)K;
            float var = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
            var = row16_sum(var);
            const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
            u32x4 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = (v[j >> 1][2 * (j & 1)] - mean) * rstd * gg[j >> 1][2 * (j & 1)] + bb[j >> 1][2 * (j & 1)];
                const float x1 = (v[j >> 1][2 * (j & 1) + 1] - mean) * rstd * gg[j >> 1][2 * (j & 1) + 1] + bb[j >> 1][2 * (j & 1) + 1];
                const unsigned q0 = ux3_pack2(x0, x1);
                const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
                const unsigned q1 = ux3_pack2(r0, r1);
                const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
                p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
            }
            uint16_t* d = xh + r * K + ((l16 ^ (r & 15)) << 3);
            *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
        }
    }
    __syncthreads();
    // ---- 2. qkv of all heads
    const int mt = wave & 3, hp = wave >> 2;
    bf16x8 xb[3][KS];
    {
        const uint16_t* xw = xh + (mt * 16 + c) * K;
#pragma unroll
        for (int sK = 0; sK < KS; ++sK)
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[p][sK] = *(const bf16x8*)(xw + p * PLANE + (((4 * sK + q) ^ c) << 3));
    }
    __syncthreads();              // every wave holds its fragments: the plane storage becomes the qkv tile
    const int grow = mt * 16 + c;                            // the lane's token row in the tile
    const int gsl = grow / S, gjt = grow - gsl * S;
    const long long gseq = seq0 + gsl;
    const bool rot = grow < ROWS && gseq < a.n_seq && gjt >= a.num_cls;
    const float2* rrow = a.rope + ((size_t)((rot ? gseq : 0) / a.times_div) * a.times_stride + (rot ? gjt - a.num_cls : 0)) * (HD / 2);
    for (int rd = 0; rd < 2; ++rd) {                         // two heads per round: wave (m-tile w & 3, head 2 rd + (w >> 2))
        {
            const int h = 2 * rd + hp;
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {                 // n-tile pairs: q, k, v of the head
                const int nt0 = jp * 8 + 2 * h;
                bf16x8 wa[2][3][KS];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int sK = 0; sK < KS; ++sK)
#pragma unroll
                        for (int p = 0; p < 3; ++p) wa[e][p][sK] = *(const bf16x8*)(a.w_qkv + ((((size_t)(nt0 + e) * KS + sK) * 3 + p) * 64 + lane) * 8);
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int sK = 0; sK < KS; ++sK)
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj)
#pragma unroll
                        for (int e = 0; e < 2; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[e][PA[jj]][sK], xb[PB[jj]][sK], acc[e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    f32x4 v = acc[e] + *(const f32x4*)(a.b_qkv + (nt0 + e) * 16 + 4 * q);
                    if (jp < 2 && rot) {                     // RoPE on q and k: dim pairs (e*16 + 4q, +1) and (+2, +3) of the head
                        const f32x4 cs = *(const f32x4*)(rrow + e * 8 + 2 * q);          // (cos, sin) of the two pairs
                        v = f32x4{v[0] * cs[0] - v[1] * cs[1], v[0] * cs[1] + v[1] * cs[0], v[2] * cs[2] - v[3] * cs[3], v[2] * cs[3] + v[3] * cs[2]};
                    }
                    *(f32x4*)(qh + grow * QS + hp * 96 + jp * 32 + e * 16 + 4 * q) = v;
                }
            }
        }
        __syncthreads();
        // ---- 3. attention: task = (sequence sl, head of the round); lane (c, q)
        for (int task = wave; task < SEQS * 2; task += 8) {
        const int sl = task >> 1, h = 2 * rd + (task & 1);
        const long long seq = seq0 + sl;
        if (seq >= a.n_seq) continue;                        // wave-uniform
        const float* base = qh + (sl * S) * QS + (task & 1) * 96;
        const float* mrow = a.mask + (size_t)(seq / a.mask_div) * S;
        // scores^T = K Q^T: A = K (row j = c, dims 8q .. 8q+7), B = Q (column i = c, the same dims); rows past the tile's last
        // sequence belong to nobody and are masked below
        const int jr = sl * S + c < BM ? c : 0;
        const f32x4 k0 = *(const f32x4*)(base + jr * QS + 32 + 8 * q), k1 = *(const f32x4*)(base + jr * QS + 32 + 8 * q + 4);
        const f32x4 q0 = *(const f32x4*)(base + jr * QS + 8 * q), q1 = *(const f32x4*)(base + jr * QS + 8 * q + 4);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[e], q0[e], sc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[e], q1[e], sc, 0, 0, 0);
        // sc[r] = q_i . k_j for query i = c, key j = 4q + r
        const bool row_ok = c < S && mrow[c < S ? c : 0] == 0.f;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * q + r;
            const bool col_ok = j < S && mrow[j < S ? j : 0] == 0.f;
            sc[r] = col_ok ? sc[r] * a.scale : -INFINITY;
            mx = sc[r] > mx ? sc[r] : mx;
        }
        { const float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; }
        { const float o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
        float pr[4], den = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = (row_ok && sc[r] > -INFINITY) ? __expf(sc[r] - mx) : 0.f; den += pr[r]; }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float inv = den > 0.f ? 1.f / den : 0.f;       // a fully masked query row yields zeros (torch SDPA semantics)
        // out = P V, k index (step s, lane group q) <-> key j = 4q + s: the A operand of step s is the lane's own pr[s]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2_ = 0; s2_ < 4; ++s2_) {
                const int j = 4 * q + s2_;
                const float vv = (j < S) ? base[j * QS + 64 + dt * 16 + c] : 0.f;
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_] * inv, vv, o, 0, 0, 0);
            }
            // o[r] = out[query 4q + r][dim dt*16 + c]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                if (i < S) a.att[(m0 + sl * S + i) * K + h * HD + dt * 16 + c] = o[r];
            }
        }
        }
        __syncthreads();              // the round's q | k | v are consumed: the next round overwrites them
    }
}

// ------------------------------------------------------------------ ALL layers of a stage in one kernel, sequences of S <= 64 tokens
// A small batch (one rally from the hub surface, the pipeline's per-clip uplift) is a dependent chain of ~80 launches of a few
// microseconds of work each, most of them one workgroup that waits on its weight fetches.  Here a workgroup of 8 waves owns
// SEQS = 64 / S whole sequences (the table stage: four 14-token sequences; the temporal / spin stages of a clip of up to 63 frames:
// one) and runs EVERY layer of the stage on them:
//   * the tokens live in registers between layers (wave w owns output features 16 w .. 16 w + 15 of all 64 rows in every GEMM, so
//     the residuals are already where the next result lands) and pass through LDS only as LayerNorm / operand staging;
//   * per layer  LN -> q | k | v of ALL heads (three 64 x 16 tiles per wave; bias and RoPE in the epilogue) -> attention on the fp32
//     matrix pipe (attention_mfma_kernel's two-pass form over ceil(S/16) key tiles, K and V read from the qkv tile in LDS) ->
//     proj + residual -> LN -> fc1 -> ReLU -> fc2 + residual: the split-bf16 arithmetic of linear_x3_kernel throughout (same
//     split, same accumulation order per output);
//   * a wave's next 12 KB weight tile (16 outputs x 128 inputs x three bf16 planes) is requested one GEMM ahead and stays in
//     flight across the LDS phases in between: the barriers wait on LDS traffic only (stage_barrier), not on the vector-memory
//     counter, which is what made the per-layer kernels (and a first version of this one: 49 us per layer for one workgroup)
//     latency-bound on a single CU's fetches.
// LDS: split planes [3][64][128] bf16 (48 KB; the attention output aliases them) | q | k tile [64][260] fp32 (65 KB; the fp32
// staging of the LayerNorms and of the MLP aliases it) | V transposed [4 heads][32][84] fp32 (42 KB: the P V operand of four keys
// is one 16-byte read; a sequence's tokens start at a multiple of 4) | 16 floats per wave = 155.5 KB.
struct StageLayerW {
    const uint16_t *w_qkv, *w_proj, *w_fc1, *w_fc2;
    const float *b_qkv, *g1, *b1, *g2, *b2, *bias1, *bias2;
};
constexpr int STAGE_MAX_LAYERS = 16;   // the layer table travels in the kernel arguments (scalar loads, pointers known to be global)
struct StageArgs {
    float* x; long long n_seq; StageLayerW layers[STAGE_MAX_LAYERS]; int n_layers;
    const float* mask; const float2* rope;
    int S, num_cls, mask_div, times_div, times_stride;
    float scale;
    // table stage without the assembled token tensor (model.py:374-378 and the gather after the stage): when `table_tok` is set, row 0
    // of sequence (b, t) is read from x[(b*T + t)] (the ball token), row 1 + n from table_tok[b*NT + n], and only row 0 is written back
    // -- to the same place.  14 of 15 token rows of the stage never exist in HBM.
    const float* table_tok; int T, NT;
    long long* stamps;                 // TTUP_STAGE_STAMPS=1: [layer][12] clock values of workgroup 0 / wave 0 at the phase boundaries (else null)
};
constexpr int STAGE_QS = 260;         // floats per row of the q | k tile: 4 heads x 64 + 4 (1040 B = 65 slots of 16 B: consecutive rows fall on consecutive slots)
constexpr int STAGE_VS = 84;          // floats per row of V^T [head][dim][token]: 4 x 84 = 16 (mod 64), so a transposed store of 4 dims x 16 tokens per lane group is conflict-free
constexpr size_t STAGE_LDS = (size_t)3 * 64 * 128 * 2 + (size_t)64 * STAGE_QS * 4 + (size_t)4 * 32 * STAGE_VS * 4 + 8 * 16 * 4;
__global__ __launch_bounds__(512) void stage_x3_kernel(StageArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t xh[];       // split planes
    constexpr int BM = 64, K = 128, PLANE = BM * K, KS = K / 32, HD = 32, QS = STAGE_QS, VS = STAGE_VS;
    float* att = (float*)xh;                                          // attention output (fp32, swizzled), while the planes are dead
    float* qh = (float*)(xh + 3 * PLANE);                             // q | k of the four heads: [row][head][q|k][32]
    float* s2 = qh;                                                   // fp32 row staging (swizzled), while the q | k tile is dead
    float* vt = qh + BM * QS;                                         // V^T: [head][dim][sequence sl at column sl*S4 + token]
    float* sinv = vt + 4 * HD * VS;
    auto swz = [&](float* b, int r, int n) __attribute__((always_inline)) { return b + r * K + ((((n >> 2) ^ (r & 15))) << 2) + (n & 3); };
    const int tid = ttup_tid_x(), lane = tid & 63, wave = tid >> 6;
    const int S = a.S, SEQS = BM / S, ROWS = SEQS * S, QT = (S + 15) >> 4, S4 = (S + 3) & ~3;
    const long long seq0 = (long long)ttup_bid_x() * SEQS;
    const long long m0 = seq0 * S, M = a.n_seq * S;
    const int q = lane >> 4, c = lane & 15;
    const int grp = tid >> 4, l16 = tid & 15;
    const int n = wave * 16 + 4 * q;                                  // the lane's four output features in every 128-wide GEMM
    auto split_store = [&](int r, int chunk, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
        u32x4 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = j < 2 ? lo[2 * j] : hi[2 * (j - 2)], x1 = j < 2 ? lo[2 * j + 1] : hi[2 * (j - 2) + 1];
            const unsigned q0 = ux3_pack2(x0, x1);
            const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
            const unsigned q1 = ux3_pack2(r0, r1);
            const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
            p0[j] = q0; p1[j] = q1; p2[j] = ux3_pack2(s0, s1);
        }
        uint16_t* d = xh + r * K + ((chunk ^ (r & 15)) << 3);
        *(u32x4*)d = p0; *(u32x4*)(d + PLANE) = p1; *(u32x4*)(d + 2 * PLANE) = p2;
    };
    // LayerNorm of row r of s2 (16 lanes per row, 8 features each) -> split planes
    auto ln_split = [&](int r, const f32x4 (&g)[2], const f32x4 (&bt)[2]) __attribute__((always_inline)) {
        f32x4 v[2] = {*(const f32x4*)swz(s2, r, 8 * l16), *(const f32x4*)swz(s2, r, 8 * l16 + 4)};
        float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
        sum = row16_sum(sum);
        const float mean = sum / (float)K;
        float var = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
        var = row16_sum(var);
        const float rstd = 1.0f / sqrtf(var / (float)K + 1e-5f);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[u][e] + bt[u][e];
        split_store(r, l16, v[0], v[1]);
    };
    // one 16-output weight tile: 4 k-steps x 3 planes, 16 bytes per lane each
    // (the scheduling barriers pin the twelve requests where they are written: left alone, the scheduler sinks them to their
    // first use -- the next GEMM -- to save registers, which is exactly the exposed latency this kernel exists to hide)
    auto load_tile = [&](const uint16_t* __restrict__ w3, int nt, bf16x8 (&w)[3][KS]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p][s] = *(const bf16x8*)(w3 + ((((size_t)nt * KS + s) * 3 + p) * 64 + lane) * 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    // acc[mt] += W_tile . planes  (64 tokens x 16 outputs x 128 inputs, six partial products smallest first)
    auto gemm = [&](const bf16x8 (&w)[3][KS], f32x4 (&acc)[4]) __attribute__((always_inline)) {
        const uint16_t* xw = xh + c * K;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 xb[3][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p][mt] = *(const bf16x8*)(xw + p * PLANE + mt * 16 * K + (((4 * s + q) ^ c) << 3));
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[PA[j]][s], xb[PB[j]][mt], acc[mt], 0, 0, 0);
        }
    };
    // ---- the tokens: global -> registers (row mt*16 + c, features n .. n+3)
    f32x4 xr[4];
    bool rot[4]; const float2* rrow[4]; int vcol[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = mt * 16 + c;
        const long long m = m0 + r;
        const int sl = r / S, jt = r - sl * S;
        const long long sq = seq0 + sl;
        const float* src = a.x + m * K;
        if (a.table_tok) src = jt == 0 ? a.x + sq * K : a.table_tok + ((sq < a.n_seq ? sq / a.T : 0) * a.NT + jt - 1) * K;
        xr[mt] = (r < ROWS && m < M) ? *(const f32x4*)(src + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        rot[mt] = r < ROWS && sq < a.n_seq && jt >= a.num_cls;
        rrow[mt] = a.rope + ((size_t)((rot[mt] ? sq : 0) / a.times_div) * a.times_stride + (rot[mt] ? jt - a.num_cls : 0)) * (HD / 2);
        vcol[mt] = r < ROWS ? sl * S4 + jt : -1;             // the row's column in V^T (rows of no sequence are not stored)
    }
    for (int i = tid; i < 4 * HD * VS + 8 * 16; i += 512) vt[i] = 0.f;          // V^T and the normalisers: never-written columns must read as finite (0 x NaN)
    // bit r: row r of the tile takes part in attention (its mask entry is 0); every wave computes the same 64 bits
    unsigned long long rowbits;
    {
        const int sl = lane / S, jt = lane - sl * S;
        const long long sq = seq0 + sl;
        rowbits = __builtin_amdgcn_ballot_w64(lane < ROWS && sq < a.n_seq && a.mask[(size_t)((lane < ROWS && sq < a.n_seq ? sq : 0) / a.mask_div) * S + jt] == 0.f);
    }
    bf16x8 wnext[3][KS];
    if (a.n_layers > 0) load_tile(a.layers[0].w_qkv, wave, wnext);
    const int hw = wave >> 1, ew = wave & 1;                 // a wave's q / k / v tile: head hw, dims 16 ew .. 16 ew + 15
    for (int li = 0; li < a.n_layers; ++li) {
        const StageLayerW& L = a.layers[li];
        auto stamp = [&](int i) __attribute__((always_inline)) { if (a.stamps && ttup_bid_x() == 0 && tid == 0) a.stamps[li * 12 + i] = (long long)__builtin_readcyclecounter(); };
        stamp(0);
        // ---- 1. LN(x) -> split planes   (small operands are requested BEFORE the weight tile that is issued next: the memory
        // counter retires in order, so waiting for them then does not wait for the tile)
        f32x4 lg[2], lb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { lg[u] = *(const f32x4*)(L.g1 + 8 * l16 + 4 * u); lb[u] = *(const f32x4*)(L.b1 + 8 * l16 + 4 * u); }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *(f32x4*)swz(s2, mt * 16 + c, n) = xr[mt];
        stage_barrier();
        ln_split(grp, lg, lb);
        ln_split(grp + 32, lg, lb);
        stage_barrier();
        stamp(1);
        // ---- 2. q | k | v: tiles wave, 8 + wave, 16 + wave of the 384 outputs (bias; RoPE on q and k)
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
            bf16x8 wc[3][KS];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
            const f32x4 b4 = *(const f32x4*)(L.b_qkv + jp * K + n);
            f32x4 cs4[4];
            if (jp < 2) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) cs4[mt] = *(const f32x4*)(rrow[mt] + ew * 8 + 2 * q);
            }
            if (jp < 2) load_tile(L.w_qkv, (jp + 1) * 8 + wave, wnext); else load_tile(L.w_proj, wave, wnext);
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm(wc, acc);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 v = acc[mt] + b4;
                if (jp < 2 && rot[mt]) {                     // dim pairs (16 ew + 4q, +1) and (+2, +3) of the head
                    const f32x4 cs = cs4[mt];
                    v = f32x4{v[0] * cs[0] - v[1] * cs[1], v[0] * cs[1] + v[1] * cs[0], v[2] * cs[2] - v[3] * cs[3], v[2] * cs[3] + v[3] * cs[2]};
                }
                if (jp < 2) *(f32x4*)(qh + (mt * 16 + c) * QS + hw * 64 + jp * 32 + ew * 16 + 4 * q) = v;
                else if (vcol[mt] >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) vt[(hw * HD + ew * 16 + 4 * q + e) * VS + vcol[mt]] = v[e];
                }
            }
        }
        stamp(2);
        stage_barrier();              // qkv complete; every wave is done with the planes: the attention output goes there
        stamp(3);
        // ---- 3. attention: task = (sequence, head, tile of 16 queries)
        for (int task = wave; task < SEQS * 4 * QT; task += 8) {
            const int qt = task % QT, sh = task / QT, h = sh & 3, sl = sh >> 2;
            const long long seq = seq0 + sl;
            if (seq >= a.n_seq) continue;                    // wave-uniform
            const float* base = qh + (sl * S) * QS + h * 64;
            const float* vbase = vt + (h * HD + c) * VS + sl * S4 + 4 * q;          // V^T[dim c][keys 4q ..] of the sequence; dims 16 + c are 16 rows on
            const int i = qt * 16 + c, ir = i < S ? i : S - 1;
            const unsigned long long seqbits = (rowbits >> (sl * S)) & (S >= 64 ? ~0ull : (1ull << S) - 1);          // bit j: key / query j of this sequence is valid
            const bool row_ok = (seqbits >> (i & 63)) & 1 && i < S;
            const unsigned long long colbits = seqbits >> (4 * q);          // bit kt*16 + r: key kt*16 + 4q + r
            const f32x4 q0 = *(const f32x4*)(base + ir * QS + 8 * q), q1 = *(const f32x4*)(base + ir * QS + 8 * q + 4);
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
            float den = 0.f;
            // NKT key tiles at once: their score chains are independent (the matrix pipe stays fed) and the scores stay in
            // registers between the maximum and the exponentials; per chain and per output the operation order is
            // attention_mfma_kernel's
            auto attend = [&](auto nkt_c) __attribute__((always_inline)) {
                constexpr int NKT = decltype(nkt_c)::value;
                f32x4 kk[NKT][2], sc[NKT], vv[NKT][2];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const int jc = kt * 16 + c, jr = jc < S ? jc : S - 1;
                    const float* kp = base + jr * QS + 32 + 8 * q;
                    kk[kt][0] = *(const f32x4*)kp; kk[kt][1] = *(const f32x4*)(kp + 4);
                    sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[kt][0][e], q0[e], sc[kt], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[kt][1][e], q1[e], sc[kt], 0, 0, 0);
                // V^T of keys kt*16 + 4q .. + 3, requested once the K fragments are dead (columns past the sequence hold other tokens,
                // zeros or -- past the array -- the normalisers: their p is 0 and all of it is finite, the storage having been cleared once)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) { vv[kt][0] = *(const f32x4*)(vbase + kt * 16); vv[kt][1] = *(const f32x4*)(vbase + kt * 16 + 16 * VS); }
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool col_ok = (colbits >> (kt * 16 + r)) & 1;
                        sc[kt][r] = col_ok ? sc[kt][r] * a.scale : -INFINITY;
                        mx = sc[kt][r] > mx ? sc[kt][r] : mx;
                    }
                { const float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; }
                { const float o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    float pr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pr[r] = (row_ok && sc[kt][r] > -INFINITY) ? __expf(sc[kt][r] - mx) : 0.f; den += pr[r]; }          // (v_exp_f32: 1 ulp; sixteen libm expf per task were a third of the attention phase)
                    // out += P V with k index (step s, lane group q) <-> key kt*16 + 4q + s: the A operand of step s is the lane's own pr[s]
#pragma unroll
                    for (int s2_ = 0; s2_ < 4; ++s2_) {
                        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vv[kt][0][s2_], o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[s2_], vv[kt][1][s2_], o1, 0, 0, 0);
                    }
                }
            };
            switch (QT) {
                case 1: attend(std::integral_constant<int, 1>{}); break;
                case 2: attend(std::integral_constant<int, 2>{}); break;
                case 3: attend(std::integral_constant<int, 3>{}); break;
                default: attend(std::integral_constant<int, 4>{}); break;
            }
            den += __shfl_xor(den, 16, 64);
            den += __shfl_xor(den, 32, 64);
            // o[r] = out[query qt*16 + 4q + r][dim c (o0) / 16 + c (o1)]: the row's 1 / den comes from the lane that owns that query
            if (q == 0) sinv[wave * 16 + c] = den > 0.f ? 1.f / den : 0.f;          // a fully masked query row yields zeros (torch SDPA semantics)
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int io = qt * 16 + 4 * q + r;
                if (io >= S) continue;
                const float inv = sinv[wave * 16 + 4 * q + r];
                *swz(att, sl * S + io, h * HD + c) = o0[r] * inv;
                *swz(att, sl * S + io, h * HD + 16 + c) = o1[r] * inv;
            }
            __builtin_amdgcn_wave_barrier();
        }
        stamp(4);
        stage_barrier();              // att complete, q | k | v consumed
        stamp(5);
        // ---- 4. att (fp32, in the plane storage) -> split planes, through registers
        {
            f32x4 t[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { t[i][0] = *(const f32x4*)swz(att, grp + 32 * i, 8 * l16); t[i][1] = *(const f32x4*)swz(att, grp + 32 * i, 8 * l16 + 4); }
            stage_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) split_store(grp + 32 * i, l16, t[i][0], t[i][1]);
        }
        stage_barrier();
        stamp(6);
        // ---- 5. x2 = proj(att) + x (stays in the lane); a copy goes to s2 for the LayerNorm
        f32x4 x2[4];
        {
            bf16x8 wc[3][KS];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
#pragma unroll
            for (int u = 0; u < 2; ++u) { lg[u] = *(const f32x4*)(L.g2 + 8 * l16 + 4 * u); lb[u] = *(const f32x4*)(L.b2 + 8 * l16 + 4 * u); }
            load_tile(L.w_fc1, wave, wnext);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) x2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm(wc, x2);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                x2[mt] += xr[mt];
                *(f32x4*)swz(s2, mt * 16 + c, n) = x2[mt];
            }
        }
        stage_barrier();
        stamp(7);
        ln_split(grp, lg, lb);
        ln_split(grp + 32, lg, lb);
        stage_barrier();
        stamp(8);
        // ---- 6. hid = relu(fc1(LN(x2)) + b1) -> s2 -> split planes
        {
            bf16x8 wc[3][KS];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
            const f32x4 b4 = *(const f32x4*)(L.bias1 + n);
            load_tile(L.w_fc2, wave, wnext);
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm(wc, acc);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 v = acc[mt] + b4;
                v = f32x4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
                *(f32x4*)swz(s2, mt * 16 + c, n) = v;
            }
        }
        stage_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = grp + 32 * i;
            split_store(r, l16, *(const f32x4*)swz(s2, r, 8 * l16), *(const f32x4*)swz(s2, r, 8 * l16 + 4));
        }
        stage_barrier();
        stamp(9);
        // ---- 7. x = fc2(hid) + b2 + x2
        {
            bf16x8 wc[3][KS];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) wc[p][s] = wnext[p][s];
            const f32x4 b4 = *(const f32x4*)(L.bias2 + n);
            if (li + 1 < a.n_layers) load_tile(a.layers[li + 1].w_qkv, wave, wnext);
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm(wc, acc);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xr[mt] = (acc[mt] + b4) + x2[mt];
        }
        stamp(10);
        stage_barrier();              // every wave is done with the planes and with s2
        stamp(11);
    }
    // ---- the tokens: registers -> global
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = mt * 16 + c;
        const long long m = m0 + r;
        if (!(r < ROWS && m < M)) continue;
        if (!a.table_tok) *(f32x4*)(a.x + m * K + n) = xr[mt];
        else {
            const int sl = r / S;
            if (r == sl * S) *(f32x4*)(a.x + (seq0 + sl) * K + n) = xr[mt];
        }
    }
}

// ------------------------------------------------------------------ token assembly helpers
// x[(b,t), 0] = ball_tok[b,t]; x[(b,t), 1+n] = table_tok[b,n]      (model.py:374-378)
__global__ void assemble_table_kernel(const float* ball_tok, const float* table_tok, float* x, int T, int NT, int D, long long total) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= total) return;
    const int d = (int)(i % D);
    long long r = i / D;
    const int n = (int)(r % (NT + 1)); r /= (NT + 1);      // r = b*T + t
    x[i] = n == 0 ? ball_tok[r * D + d] : table_tok[((r / T) * NT + (n - 1)) * D + d];
}
// y[r] = x[r*stride_tok] rows (token 0 of every sequence)            (model.py:383-384)
__global__ void gather_rows_kernel(const float* x, float* y, int D, int seq_tokens, long long total) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= total) return;
    const int d = (int)(i % D);
    const long long r = i / D;
    y[i] = x[(r * seq_tokens) * D + d];
}
// y[b, 0] = cls; y[b, 1+t] = x[b, t]                                 (model.py:560)
__global__ void prepend_cls_kernel(const float* x, const float* cls, float* y, int T, int D, long long total) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (i >= total) return;
    const int d = (int)(i % D);
    long long r = i / D;
    const int t = (int)(r % (T + 1)); const long long b = r / (T + 1);
    y[i] = t == 0 ? cls[d] : x[(b * T + (t - 1)) * D + d];
}
// masks: mask (B,T) {0,1} -> additive m1 (B,T), m2 (B,T+1) with leading 0; table (B,13,3) -> tmask (B,14), txy (B*13,2)
__global__ void prepare_kernel(const float* mask, const float* table, float* m1, float* m2, float* tmask, float* txy, int B, int T, int NT, int* flags) {
    const long long i = (long long)ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    const long long nmask = (long long)B * T, ntab = (long long)B * NT;
    int fl = 0;
    if (i < nmask) {
        const float m = mask[i];
        const float add = m == 0.f ? -INFINITY : 0.f;
        m1[i] = add;
        const long long b = i / T; const int t = (int)(i % T);
        m2[b * (T + 1) + 1 + t] = add;
        if (t == 0) m2[b * (T + 1)] = 0.f;
        // bit0: some m==0, bit1: some m==1, bit2: some m<0, bit3: some m>1  (min==0 && max==1  <=>  flags==3)
        fl = m == 0.f ? 1 : m == 1.f ? 2 : m < 0.f ? 4 : 8;
    } else if (i < nmask + ntab) {
        const long long j = i - nmask;
        const long long b = j / NT; const int n = (int)(j % NT);
        tmask[b * (NT + 1) + 1 + n] = table[j * 3 + 2] == 1.f ? 0.f : -INFINITY;      // KEYPOINT_VISIBLE == 1, model.py:363
        if (n == 0) tmask[b * (NT + 1)] = 0.f;
        txy[j * 2] = table[j * 3]; txy[j * 2 + 1] = table[j * 3 + 1];
    }
    // one atomic per wave (every thread used to hit the one flag word: 1.4 ms per call at B = 10 000)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fl |= __shfl_xor(fl, off, 64);
    if ((ttup_tid_x() & 63) == 0 && fl) atomicOr(flags, fl);
}
__global__ void rotationaxes_kernel(const float* rot, const float* pos, int B, int T, float* out) {
    const int b = ttup_bid_x() * ttup_bdim_x() + ttup_tid_x();
    if (b >= B) return;
    const float* p = pos + (size_t)b * T * 3;
    const float vx = p[3] - p[0], vy = p[4] - p[1];
    const float nrm = sqrtf(vx * vx + vy * vy);
    const float ex = vx / nrm, ey = vy / nrm;            // e_x = (ex, ey, 0); e_y = e_z x e_x = (-ey, ex, 0)
    const float* r = rot + (size_t)b * 3;
    out[b * 3 + 0] = r[0] * ex + r[1] * ey + r[2] * 0.f;
    out[b * 3 + 1] = r[0] * (-ey) + r[1] * ex + r[2] * 0.f;
    out[b * 3 + 2] = r[0] * 0.f + r[1] * 0.f + r[2] * 1.f;
}

struct Layer { Linear qkv, proj, fc1, fc2; float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; };
struct Mlp2 { Linear fc1, fc2; };
struct Head { Linear fc1, fc2, fc3; };

}  // namespace

struct ttup_uplift {
    int D = 0, heads = 0, hd = 0, n_table = 13, max_batch = 0, max_len = 0, chunk = 1;
    std::vector<Layer> pos_layers, layers, second;
    Mlp2 ball_embed, table_embed;
    Head position_head, rotation_head;
    float* cls_dev = nullptr; float* inv_freq_dev = nullptr; float* table_times_dev = nullptr;
    std::vector<StageLayerW> stage_pos, stage_first, stage_second;      // weight pointers of the three stages' layers (stage_x3_kernel); empty = not available
    long long stage_launches = 0;
    float2 *rope = nullptr, *table_rope = nullptr;      // (cos, sin) tables: [chunk*max_len][hd/2] per forward, [n_table][hd/2] fixed
    std::vector<void*> allocs;
    // scratch (sized for `chunk` trajectories of max_len tokens)
    float *x = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *x2 = nullptr, *tok = nullptr, *ttok = nullptr, *h1 = nullptr;
    float *m1 = nullptr, *m2 = nullptr, *tmask = nullptr, *txy = nullptr, *tmp_small = nullptr;
    int* flags_dev = nullptr;
    // Small batches (a rally or a handful of them: the hub surface, the pipeline's per-clip uplift) are launch-bound -- about
    // eighty kernels of a few microseconds each.  Their forward is captured once per (batch, length) into a hipGraph that works
    // on handle-owned input / output buffers and is replayed with one launch (+ six small copies around it).
    struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; };
    std::map<std::pair<int, int>, GraphEntry> graphs;
    bool graphs_off = false;
    float *g_ball = nullptr, *g_table = nullptr, *g_mask = nullptr, *g_times = nullptr, *g_rot = nullptr, *g_pos = nullptr;
    long long graph_tokens = 0;          // largest batch * len served by a graph
    long long graph_replays = 0;
    ~ttup_uplift() {
        for (auto& kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        for (void* p : allocs) if (p) (void)hipFree(p);
    }
};

namespace {

struct Reader {
    const char* p; size_t left;
    bool take(std::vector<float>* v, size_t expect) {
        int n;
        if (left < 4) return false;
        memcpy(&n, p, 4); p += 4; left -= 4;
        if ((size_t)n != expect || left < expect * 4) return false;
        v->resize(expect); memcpy(v->data(), p, expect * 4); p += expect * 4; left -= expect * 4;
        return true;
    }
};

int dev_copy(ttup_uplift* net, const std::vector<float>& v, float** out) {
    void* d = nullptr;
    TTUP_HIP_CHECK(hipMalloc(&d, v.size() * 4 + 16));
    net->allocs.push_back(d);
    TTUP_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    *out = (float*)d;
    return TTUP_OK;
}
int dev_alloc(ttup_uplift* net, size_t n_floats, float** out) {
    void* d = nullptr;
    TTUP_HIP_CHECK(hipMalloc(&d, n_floats * 4 + 16));
    net->allocs.push_back(d);
    *out = (float*)d;
    return TTUP_OK;
}

int make_linear(ttup_uplift* net, Reader& r, int n, int k, bool has_bias, Linear* L) {
    std::vector<float> w, b;
    TTUP_REQUIRE(r.take(&w, (size_t)n * k), TTUP_EFORMAT, "uplift blob: bad weight record (%dx%d)", n, k);
    if (has_bias) TTUP_REQUIRE(r.take(&b, n), TTUP_EFORMAT, "uplift blob: bad bias record (%d)", n);
    L->n = n; L->k = k; L->mfma = (k % 16 == 0);
    int rc;
    if (L->mfma) {
        const int ntiles = (n + 15) / 16, ks4 = k / 16, kq = k / 4;
        std::vector<float> p((size_t)ntiles * ks4 * 64 * 4, 0.f);
        for (int nt = 0; nt < ntiles; ++nt)
            for (int s4 = 0; s4 < ks4; ++s4)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int row = nt * 16 + (l & 15), kk = (l >> 4) * kq + s4 * 4 + j;
                        p[(((size_t)nt * ks4 + s4) * 64 + l) * 4 + j] = row < n ? w[(size_t)row * k + kk] : 0.f;
                    }
        rc = dev_copy(net, p, &L->w_dev);
        if (rc == TTUP_OK && k == 128 && n % 4 == 0) {
            const int ks = k / 32;
            std::vector<uint16_t> p3((size_t)ntiles * ks * 3 * 64 * 8, 0);
            for (int nt = 0; nt < ntiles; ++nt)
                for (int s_ = 0; s_ < ks; ++s_)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = nt * 16 + (l & 15), kk = s_ * 32 + (l >> 4) * 8 + j;
                            const float v = row < n ? w[(size_t)row * k + kk] : 0.f;
                            const bf16_t a0 = f32_to_bf16(v);
                            const float r1 = v - bf16_to_f32(a0);
                            const bf16_t a1 = f32_to_bf16(r1);
                            const bf16_t a2 = f32_to_bf16(r1 - bf16_to_f32(a1));
                            const size_t base = (((size_t)nt * ks + s_) * 3) * 512 + (size_t)l * 8 + j;
                            p3[base] = a0; p3[base + 512] = a1; p3[base + 1024] = a2;
                        }
            void* d = nullptr;
            TTUP_HIP_CHECK(hipMalloc(&d, p3.size() * 2));
            net->allocs.push_back(d);
            TTUP_HIP_CHECK(hipMemcpy(d, p3.data(), p3.size() * 2, hipMemcpyHostToDevice));
            L->w3_dev = (uint16_t*)d;
        }
    } else rc = dev_copy(net, w, &L->w_dev);
    if (rc) return rc;
    if (has_bias) { rc = dev_copy(net, b, &L->b_dev); if (rc) return rc; }
    return TTUP_OK;
}

int make_vec(ttup_uplift* net, Reader& r, int n, float** out) {
    std::vector<float> v;
    TTUP_REQUIRE(r.take(&v, n), TTUP_EFORMAT, "uplift blob: bad vector record (%d)", n);
    return dev_copy(net, v, out);
}

int make_layer(ttup_uplift* net, Reader& r, Layer* L) {
    const int D = net->D;
    int rc;
    if ((rc = make_linear(net, r, 3 * D, D, true, &L->qkv))) return rc;
    if ((rc = make_linear(net, r, D, D, false, &L->proj))) return rc;      // no bias: model.py:268 / :162
    if ((rc = make_linear(net, r, D, D, true, &L->fc1))) return rc;
    if ((rc = make_linear(net, r, D, D, true, &L->fc2))) return rc;
    if ((rc = make_vec(net, r, D, &L->g1))) return rc;
    if ((rc = make_vec(net, r, D, &L->b1))) return rc;
    if ((rc = make_vec(net, r, D, &L->g2))) return rc;
    if ((rc = make_vec(net, r, D, &L->b2))) return rc;
    return TTUP_OK;
}
int make_mlp2(ttup_uplift* net, Reader& r, int din, Mlp2* m) {
    int rc;
    if ((rc = make_linear(net, r, net->D, din, true, &m->fc1))) return rc;
    return make_linear(net, r, net->D, net->D, true, &m->fc2);
}
int make_head(ttup_uplift* net, Reader& r, Head* h) {
    const int D = net->D;
    int rc;
    if ((rc = make_linear(net, r, D / 2, D, true, &h->fc1))) return rc;
    if ((rc = make_linear(net, r, D / 4, D / 2, true, &h->fc2))) return rc;
    return make_linear(net, r, 3, D / 4, true, &h->fc3);
}

int run_linear(const Linear& L, const float* x, int ldx, long long M, const float* gamma, const float* beta, int relu,
               const float* res, int ldr, float* out, int ldo, hipStream_t st) {
    if (M == 0) return TTUP_OK;
    if (!L.mfma) {
        TTUP_REQUIRE(!gamma && !res, TTUP_EINVAL, "small linear: LN/residual unsupported");
        const long long total = M * L.n;
        hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, L.w_dev, L.b_dev, out, ldo, M, L.n, L.k, relu);
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    LinArgs a;
    a.x = x; a.ldx = ldx; a.w = L.w_dev; a.bias = L.b_dev; a.gamma = gamma; a.beta = beta; a.res = res; a.ldr = ldr;
    a.out = out; a.ldo = ldo; a.M = (int)M; a.N = L.n; a.K = L.k; a.relu = relu;
    TTUP_REQUIRE(ldx % 4 == 0 && L.k <= 256, TTUP_EINVAL, "linear: row stride %d / K %d unsupported", ldx, L.k);
    // 128-row tiles once there are enough rows to fill the chip twice over, 64-row tiles below that
    const bool big = M >= 128 * 512;
    const int ntw = L.n > 128 ? 3 : L.n > 64 ? 2 : 1;
    const int bm = big ? 128 : 64;
    const dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((L.n + 64 * ntw - 1) / (64 * ntw)));
    static const bool exact = getenv("TTUP_F32_EXACT") != nullptr;
    if (L.w3_dev && !exact && ldo % 4 == 0 && (!res || ldr % 4 == 0)) {
        const size_t smem3 = (size_t)3 * bm * 128 * sizeof(uint16_t);
#define TTUP_LIN3(LN_, NTW_, MH_)                                                                                             \
    do {                                                                                                                      \
        if (int rc_ = ensure_max_lds((const void*)linear_x3_kernel<LN_, NTW_, MH_>, 160 * 1024)) return rc_;                  \
        hipLaunchKernelGGL((linear_x3_kernel<LN_, NTW_, MH_>), grid, dim3(256 * MH_), smem3, st, a, (const uint16_t*)L.w3_dev); \
    } while (0)
#define TTUP_LIN3_N(LN_, MH_)                                         \
    do {                                                              \
        if (ntw == 3) TTUP_LIN3(LN_, 3, MH_);                         \
        else if (ntw == 2) TTUP_LIN3(LN_, 2, MH_);                    \
        else TTUP_LIN3(LN_, 1, MH_);                                  \
    } while (0)
        if (gamma) { if (big) TTUP_LIN3_N(true, 2); else TTUP_LIN3_N(true, 1); }
        else { if (big) TTUP_LIN3_N(false, 2); else TTUP_LIN3_N(false, 1); }
#undef TTUP_LIN3_N
#undef TTUP_LIN3
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    const size_t smem = (size_t)4 * bm * (L.k / 4 + 4) * sizeof(float);
#define TTUP_LIN(LN_, NTW_, MH_)                                                                                              \
    do {                                                                                                                      \
        if (int rc_ = ensure_max_lds((const void*)linear_kernel<LN_, NTW_, MH_>, 160 * 1024)) return rc_;                     \
        hipLaunchKernelGGL((linear_kernel<LN_, NTW_, MH_>), grid, dim3(256 * MH_), smem, st, a);                              \
    } while (0)
#define TTUP_LIN_N(LN_, MH_)                                          \
    do {                                                              \
        if (ntw == 3) TTUP_LIN(LN_, 3, MH_);                          \
        else if (ntw == 2) TTUP_LIN(LN_, 2, MH_);                     \
        else TTUP_LIN(LN_, 1, MH_);                                   \
    } while (0)
    if (gamma) { if (big) TTUP_LIN_N(true, 2); else TTUP_LIN_N(true, 1); }
    else { if (big) TTUP_LIN_N(false, 2); else TTUP_LIN_N(false, 1); }
#undef TTUP_LIN_N
#undef TTUP_LIN
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

template <int HD>
void launch_attention(const AttnArgs& a, hipStream_t st) {
    const int S = a.S;
    const int P = S <= 16 ? 16 : S <= 32 ? 32 : S <= 64 ? 64 : 128;
    const int threads = P == 128 ? 128 : 64, G = threads / P;
    const size_t smem = ((size_t)G * (2 * S * HD + 16) + (size_t)G * S) * sizeof(float);
    const dim3 grid((unsigned)((a.n_seq + G - 1) / G), a.heads);
    switch (P) {
        case 16: hipLaunchKernelGGL((attention_kernel<HD, 16>), grid, dim3(threads), smem, st, a); break;
        case 32: hipLaunchKernelGGL((attention_kernel<HD, 32>), grid, dim3(threads), smem, st, a); break;
        case 64: hipLaunchKernelGGL((attention_kernel<HD, 64>), grid, dim3(threads), smem, st, a); break;
        default: hipLaunchKernelGGL((attention_kernel<HD, 128>), grid, dim3(threads), smem, st, a); break;
    }
}

int run_attention(ttup_uplift* net, const float* qkv, float* out, int n_seq, int S, int num_cls, const float* mask, int mask_div,
                  const float2* rope, int times_div, int times_stride, hipStream_t st) {
    AttnArgs a;
    a.qkv = qkv; a.out = out; a.mask = mask; a.rope = rope;
    a.n_seq = n_seq; a.S = S; a.D = net->D; a.heads = net->heads; a.hd = net->hd; a.num_cls = num_cls;
    a.mask_div = mask_div; a.times_div = times_div; a.times_stride = times_stride;
    a.scale = 1.0f / sqrtf((float)net->hd);
    static const bool scalar_attn = getenv("TTUP_UPLIFT_SCALAR_ATTENTION") != nullptr || getenv("TTUP_F32_EXACT") != nullptr;
    if (net->hd == 32 && net->D == 128 && S > 16 && S <= 512 && !scalar_attn) {
        AttnMArgs m;
        m.qkv = qkv; m.out = out; m.mask = mask; m.rope = rope; m.n_seq = n_seq; m.S = S; m.num_cls = num_cls;
        m.mask_div = mask_div; m.times_div = times_div; m.times_stride = times_stride; m.scale = a.scale;
        const int KT = (S + 15) / 16;
        static const bool two_pass = getenv("TTUP_UPLIFT_ATTENTION_2PASS") != nullptr;          // the first matrix-pipe form (cross-check)
        if (KT <= 8 && !two_pass) {
            const size_t smem8 = ((size_t)KT * 16 * ATTM_KS + (size_t)32 * (KT * 16 + 4) + 64) * sizeof(float);
            if (KT <= 4) hipLaunchKernelGGL(attention_mfma8_kernel<1>, dim3((KT + 3) / 4, net->heads, n_seq), dim3(256), smem8, st, m);
            else hipLaunchKernelGGL(attention_mfma8_kernel<2>, dim3((KT + 3) / 4, net->heads, n_seq), dim3(256), smem8, st, m);
            TTUP_LAUNCH_CHECK();
            return TTUP_OK;
        }
        const size_t smem = ((size_t)2 * KT * 16 * ATTM_KS + 64) * sizeof(float);
        if (int rc = ensure_max_lds((const void*)attention_mfma_kernel, 160 * 1024)) return rc;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3((KT + 3) / 4, net->heads, n_seq), dim3(256), smem, st, m);
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    TTUP_REQUIRE(((size_t)2 * S * net->hd + 16 + S) * sizeof(float) <= 64 * 1024, TTUP_EINVAL, "attention: sequence length %d too long", S);
    switch (net->hd) {
        case 8: launch_attention<8>(a, st); break;
        case 16: launch_attention<16>(a, st); break;
        case 24: launch_attention<24>(a, st); break;
        case 32: launch_attention<32>(a, st); break;
        default: set_error("attention: head_dim %d unsupported", net->hd); return TTUP_EINVAL;
    }
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// SimpleStaticLayer.forward (model.py:278-300) on x [n_seq*S][D] in place (x2 is scratch of the same size)
int run_layer(ttup_uplift* net, const Layer& L, float* x, long long tokens, int n_seq, int S, int num_cls,
              const float* mask, int mask_div, const float2* rope, int times_div, int times_stride, hipStream_t st) {
    const int D = net->D;
    int rc;
    static const bool exact0 = getenv("TTUP_F32_EXACT") != nullptr, unfused0 = getenv("TTUP_UPLIFT_UNFUSED") != nullptr;
    if (D == 128 && net->heads == 4 && S <= 16 && L.qkv.w3_dev && !exact0 && !unfused0) {
        // short sequences (table stage): LN + qkv + RoPE + attention in one kernel, qkv never leaves the CU (attn_block_x3_kernel)
        AttnBlockArgs a;
        a.x = x; a.att = net->att; a.n_seq = n_seq; a.w_qkv = L.qkv.w3_dev; a.b_qkv = L.qkv.b_dev; a.g1 = L.g1; a.b1 = L.b1;
        a.mask = mask; a.rope = rope; a.S = S; a.num_cls = num_cls; a.mask_div = mask_div; a.times_div = times_div; a.times_stride = times_stride;
        a.scale = 1.0f / sqrtf((float)net->hd);
        const int seqs = 64 / S;
        const size_t smem = (size_t)64 * ATTN_QS * sizeof(float);          // (>= the 48 KB of the three split planes it first holds)
        const dim3 grid((unsigned)((n_seq + seqs - 1) / seqs));
        if ((rc = ensure_max_lds((const void*)attn_block_x3_kernel, 160 * 1024))) return rc;
        hipLaunchKernelGGL(attn_block_x3_kernel, grid, dim3(512), smem, st, a);
        TTUP_LAUNCH_CHECK();
    } else {
        static const bool qkv_linear = getenv("TTUP_UPLIFT_QKV_LINEAR") != nullptr;          // the general linear kernel instead (cross-check)
        // (small launches only: on a full device the general kernel -- 128-token tiles, two workgroups per 128 x 384 block -- is 4 % ahead,
        // B = 10 000: 65.2 k vs 62.8 k trajectories/s; three 121-token trajectories: 0.712 -> 0.689 ms with this one)
        if (D == 128 && L.qkv.w3_dev && L.qkv.n == 384 && !exact0 && !unfused0 && !qkv_linear && tokens <= 64 * 256) {
            QkvArgs qa{x, net->qkv, tokens, L.qkv.w3_dev, L.qkv.b_dev, L.g1, L.b1};
            hipLaunchKernelGGL(qkv_block8_x3_kernel, dim3((unsigned)((tokens + 63) / 64)), dim3(512), (size_t)3 * 64 * 128 * sizeof(uint16_t), st, qa);
            TTUP_LAUNCH_CHECK();
        } else if ((rc = run_linear(L.qkv, x, D, tokens, L.g1, L.b1, 0, nullptr, 0, net->qkv, 3 * D, st))) return rc;
        if ((rc = run_attention(net, net->qkv, net->att, n_seq, S, num_cls, mask, mask_div, rope, times_div, times_stride, st))) return rc;
    }
    static const bool exact = getenv("TTUP_F32_EXACT") != nullptr, unfused = getenv("TTUP_UPLIFT_UNFUSED") != nullptr;
    if (D == 128 && L.proj.w3_dev && L.fc1.w3_dev && L.fc2.w3_dev && !exact && !unfused) {
        // x = fc2(relu(fc1(LN(proj(att) + x)))) + (proj(att) + x) in one pass over the tokens (mlp_block_x3_kernel)
        MlpArgs a;
        a.att = net->att; a.x = x; a.M = tokens;
        a.w_proj = L.proj.w3_dev; a.w_fc1 = L.fc1.w3_dev; a.w_fc2 = L.fc2.w3_dev;
        a.g2 = L.g2; a.b2 = L.b2; a.bias1 = L.fc1.b_dev; a.bias2 = L.fc2.b_dev;
        // 64-token tiles (80 KB of LDS: two workgroups per CU) also for large token counts: 2 % faster at B = 10 000 than the 128-token
        // tile (160 KB, one workgroup per CU) although every tile then streams the weights again; TTUP_UPLIFT_MLP_BM128=1 selects the latter
        static const bool bm128 = getenv("TTUP_UPLIFT_MLP_BM128") != nullptr;
        const bool big = tokens >= 128 * 512 && bm128;
        const int bm = big ? 128 : 64;
        const size_t smem = (size_t)3 * bm * 128 * sizeof(uint16_t) + (size_t)bm * 128 * sizeof(float);
        const dim3 grid((unsigned)((tokens + bm - 1) / bm));
        if (big) {
            if ((rc = ensure_max_lds((const void*)mlp_block_x3_kernel<2>, 160 * 1024))) return rc;
            hipLaunchKernelGGL(mlp_block_x3_kernel<2>, grid, dim3(512), smem, st, a);
        } else {
            static const bool four = getenv("TTUP_UPLIFT_MLP_4WAVES") != nullptr;          // the round-3 form: 4 waves, two n-tiles each
            if (four) {
                if ((rc = ensure_max_lds((const void*)mlp_block_x3_kernel<1>, 160 * 1024))) return rc;
                hipLaunchKernelGGL(mlp_block_x3_kernel<1>, grid, dim3(256), smem, st, a);
            } else {
                if ((rc = ensure_max_lds((const void*)mlp_block8_x3_kernel, 160 * 1024))) return rc;
                hipLaunchKernelGGL(mlp_block8_x3_kernel, grid, dim3(512), smem, st, a);
            }
        }
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    if ((rc = run_linear(L.proj, net->att, D, tokens, nullptr, nullptr, 0, x, D, net->x2, D, st))) return rc;       // x2 = proj(att) + x
    if ((rc = run_linear(L.fc1, net->x2, D, tokens, L.g2, L.b2, 1, nullptr, 0, net->hid, D, st))) return rc;          // hid = relu(fc1(LN(x2)))
    return run_linear(L.fc2, net->hid, D, tokens, nullptr, nullptr, 0, net->x2, D, x, D, st);                       // x = fc2(hid) + x2
}

// the layers' weight pointers for stage_x3_kernel (left empty when a layer has no split-bf16 image or the table would not fit)
void make_stage(ttup_uplift* net, const std::vector<Layer>& layers, std::vector<StageLayerW>* out) {
    out->clear();
    if (layers.empty() || layers.size() > (size_t)STAGE_MAX_LAYERS || net->D != 128 || net->heads != 4) return;
    for (const Layer& L : layers) {
        if (!L.qkv.w3_dev || !L.proj.w3_dev || !L.fc1.w3_dev || !L.fc2.w3_dev) { out->clear(); return; }
        out->push_back(StageLayerW{L.qkv.w3_dev, L.proj.w3_dev, L.fc1.w3_dev, L.fc2.w3_dev, L.qkv.b_dev, L.g1, L.b1, L.g2, L.b2, L.fc1.b_dev, L.fc2.b_dev});
    }
}

// Every layer of a stage: one stage_x3_kernel launch when the sequences fit a 64-token tile (the table stage always; the temporal
// and spin stages of clips of up to 63 frames), else layer by layer.  The kernel holds 156 KB of LDS -- one workgroup per CU -- and
// still beats the per-layer kernels (two per CU) on a full device: 55 k cycles per 64-token layer against 19 k (attention block,
// bound by the L1 traffic of its weight fragments: every m-tile wave streams its head's weights) + 38 k (MLP block); B = 10 000,
// T = 120: 50.1 k -> 55.4 k trajectories/s, B = 4096, T = 50: 123 k -> 149 k.  TTUP_UPLIFT_STAGE_WG caps the launch size it is used for.
int run_stage(ttup_uplift* net, const std::vector<Layer>& layers, const std::vector<StageLayerW>& stage, float* x, long long tokens, int n_seq, int S, int num_cls,
              const float* mask, int mask_div, const float2* rope, int times_div, int times_stride, hipStream_t st,
              const float* table_tok = nullptr, int T = 0, int NT = 0, bool* fused_tokens = nullptr) {
    // (table_tok: the table stage.  When the stage kernel runs, `x` is then the ball-token tensor [n_seq][D], read and written in
    // place, and *fused_tokens = true; otherwise the caller assembles / gathers around the per-layer kernels, which get `x` as usual)
    if (fused_tokens) *fused_tokens = false;
    static const bool off = getenv("TTUP_F32_EXACT") != nullptr || getenv("TTUP_UPLIFT_UNFUSED") != nullptr || getenv("TTUP_UPLIFT_NO_STAGE") != nullptr;
    static const long long max_wg = getenv("TTUP_UPLIFT_STAGE_WG") ? atoll(getenv("TTUP_UPLIFT_STAGE_WG")) : (1ll << 40);
    if (!stage.empty() && !off && S <= 64 && n_seq > 0 && (64 / S) * ((S + 3) & ~3) <= STAGE_VS) {          // (V^T holds every sequence of the tile at a multiple of 4)
        const int seqs = 64 / S;
        const long long wgs = ((long long)n_seq + seqs - 1) / seqs;
        if (wgs <= max_wg) {
            StageArgs a;
            a.x = x; a.n_seq = n_seq; a.n_layers = (int)layers.size();
            memcpy(a.layers, stage.data(), stage.size() * sizeof(StageLayerW));
            a.mask = mask; a.rope = rope; a.S = S; a.num_cls = num_cls; a.mask_div = mask_div; a.times_div = times_div; a.times_stride = times_stride;
            a.scale = 1.0f / sqrtf((float)net->hd);
            a.table_tok = table_tok; a.T = T; a.NT = NT;
            if (fused_tokens) *fused_tokens = table_tok != nullptr;
            if (int rc = ensure_max_lds((const void*)stage_x3_kernel, 160 * 1024)) return rc;
            static const bool want_stamps = getenv("TTUP_STAGE_STAMPS") != nullptr;
            static long long* stamps_dev = nullptr;
            a.stamps = nullptr;
            if (want_stamps) {
                if (!stamps_dev) TTUP_HIP_CHECK(hipMalloc((void**)&stamps_dev, STAGE_MAX_LAYERS * 12 * sizeof(long long)));
                a.stamps = stamps_dev;
            }
            hipLaunchKernelGGL(stage_x3_kernel, dim3((unsigned)wgs), dim3(512), STAGE_LDS, st, a);
            TTUP_LAUNCH_CHECK();
            if (want_stamps) {          // debugging aid (synchronises): cycles between the phase boundaries of the LAST layer, wave 0 of workgroup 0
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(st, &cs);
                if (cs == hipStreamCaptureStatusNone) {
                    std::vector<long long> h((size_t)a.n_layers * 12);
                    TTUP_HIP_CHECK(hipStreamSynchronize(st));
                    TTUP_HIP_CHECK(hipMemcpy(h.data(), stamps_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
                    const long long* t = h.data() + (size_t)(a.n_layers - 1) * 12;
                    fprintf(stderr, "stage S=%d n_seq=%d wgs=%lld layers=%d: whole stage %lld clk; last layer: ln1 %lld qkv %lld wait %lld attn %lld wait %lld att->planes %lld proj %lld ln2 %lld fc1+split %lld fc2 %lld wait %lld\n",
                            S, n_seq, wgs, a.n_layers, h[(size_t)(a.n_layers - 1) * 12 + 11] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5],
                            t[7] - t[6], t[8] - t[7], t[9] - t[8], t[10] - t[9], t[11] - t[10]);
                }
            }
            net->stage_launches++;
            return TTUP_OK;
        }
    }
    if (table_tok) return TTUP_OK;          // declined (*fused_tokens is false, nothing launched): the caller assembles the tokens and calls again
    for (const Layer& L : layers)
        if (int rc = run_layer(net, L, x, tokens, n_seq, S, num_cls, mask, mask_div, rope, times_div, times_stride, st)) return rc;
    return TTUP_OK;
}

int run_head(ttup_uplift* net, const Head& h, const float* x, int ldx, long long M, float* out, hipStream_t st) {
    const int D = net->D;
    int rc;
    if ((rc = run_linear(h.fc1, x, ldx, M, nullptr, nullptr, 1, nullptr, 0, net->hid, D / 2, st))) return rc;
    if ((rc = run_linear(h.fc2, net->hid, D / 2, M, nullptr, nullptr, 1, nullptr, 0, net->att, D / 4, st))) return rc;
    return run_linear(h.fc3, net->att, D / 4, M, nullptr, nullptr, 0, nullptr, 0, out, 3, st);
}

int forward_chunk(ttup_uplift* net, const float* ball, const float* table, const float* mask, const float* times, int B, int T,
                  float* rot, float* pos, hipStream_t st) {
    const int D = net->D, NT = net->n_table, S1 = NT + 1;
    int rc;
    {
        const long long n = (long long)B * T + (long long)B * NT;
        hipLaunchKernelGGL(prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, table, net->m1, net->m2, net->tmask, net->txy, B, T, NT, net->flags_dev);
        TTUP_LAUNCH_CHECK();
    }
    {
        const long long n = (long long)B * T * (net->hd / 2);
        hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, times, net->inv_freq_dev, net->rope, net->hd / 2, n);
        TTUP_LAUNCH_CHECK();
    }
    // embeddings
    if ((rc = run_linear(net->ball_embed.fc1, ball, 2, (long long)B * T, nullptr, nullptr, 1, nullptr, 0, net->h1, D, st))) return rc;
    if ((rc = run_linear(net->ball_embed.fc2, net->h1, D, (long long)B * T, nullptr, nullptr, 0, nullptr, 0, net->tok, D, st))) return rc;
    if ((rc = run_linear(net->table_embed.fc1, net->txy, 2, (long long)B * NT, nullptr, nullptr, 1, nullptr, 0, net->h1, D, st))) return rc;
    if ((rc = run_linear(net->table_embed.fc2, net->h1, D, (long long)B * NT, nullptr, nullptr, 0, nullptr, 0, net->ttok, D, st))) return rc;
    // table stage: every (b, t) is a 14-token sequence [ball token, 13 table tokens]; its row 0 replaces the ball token afterwards
    const long long tok1 = (long long)B * T * S1;
    static const bool no_token_fusion = getenv("TTUP_UPLIFT_ASSEMBLE") != nullptr;
    bool fused = false;
    if (!no_token_fusion) {
        // stage kernel: reads the two token tensors itself and writes row 0 only (nothing has been launched if it declines)
        if ((rc = run_stage(net, net->pos_layers, net->stage_pos, net->tok, tok1, B * T, S1, 1, net->tmask, T, net->table_rope, 1, 0, st, net->ttok, T, NT, &fused))) return rc;
    }
    if (!fused) {
        const long long total = tok1 * D;
        hipLaunchKernelGGL(assemble_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, net->tok, net->ttok, net->x, T, NT, D, total);
        TTUP_LAUNCH_CHECK();
        if ((rc = run_stage(net, net->pos_layers, net->stage_pos, net->x, tok1, B * T, S1, 1, net->tmask, T, net->table_rope, 1, 0, st))) return rc;
        const long long total2 = (long long)B * T * D;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, st, net->x, net->tok, D, S1, total2);
        TTUP_LAUNCH_CHECK();
    }
    // temporal stage (tok is [B*T][D])
    if ((rc = run_stage(net, net->layers, net->stage_first, net->tok, (long long)B * T, B, T, 0, net->m1, 1, net->rope, 1, T, st))) return rc;
    if ((rc = run_head(net, net->position_head, net->tok, D, (long long)B * T, pos, st))) return rc;
    // spin stage
    {
        const long long total = (long long)B * (T + 1) * D;
        hipLaunchKernelGGL(prepend_cls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, net->tok, net->cls_dev, net->x, T, D, total);
        TTUP_LAUNCH_CHECK();
    }
    if ((rc = run_stage(net, net->second, net->stage_second, net->x, (long long)B * (T + 1), B, T + 1, 1, net->m2, 1, net->rope, 1, T, st))) return rc;
    // rotation head on the cls rows (row stride (T+1)*D)
    return run_head(net, net->rotation_head, net->x, (T + 1) * D, B, rot, st);
}

}  // namespace

extern "C" int ttup_uplift_create(const void* blob, size_t blob_bytes, int max_batch, int max_len, ttup_uplift** out) {
    TTUP_REQUIRE(blob && out, TTUP_EINVAL, "ttup_uplift_create: null pointer");
    TTUP_REQUIRE(max_batch > 0 && max_len > 0, TTUP_EINVAL, "ttup_uplift_create: max_batch and max_len must be positive");
    TTUP_REQUIRE(blob_bytes >= 40 && memcmp(blob, "TTUPUPL1", 8) == 0, TTUP_EFORMAT, "uplift blob: bad magic");
    int ndev = 0;
    TTUP_HIP_CHECK(hipGetDeviceCount(&ndev));
    TTUP_REQUIRE(ndev > 0, TTUP_EHIP, "ttup_uplift_create: no HIP device");
    int hdr[8];
    memcpy(hdr, (const char*)blob + 8, sizeof hdr);
    std::unique_ptr<ttup_uplift> net(new ttup_uplift);
    net->D = hdr[0]; net->heads = hdr[1]; net->n_table = hdr[5];
    const int n_pos = hdr[2], n_first = hdr[3], n_second = hdr[4];
    TTUP_REQUIRE(net->D > 0 && net->D % 32 == 0 && net->D <= 256 && net->heads > 0 && net->D % net->heads == 0, TTUP_EFORMAT,
                 "uplift blob: dim %d / heads %d unsupported", net->D, net->heads);
    net->hd = net->D / net->heads;
    TTUP_REQUIRE(net->hd == 8 || net->hd == 16 || net->hd == 24 || net->hd == 32, TTUP_EFORMAT, "uplift blob: head_dim %d unsupported", net->hd);
    TTUP_REQUIRE(net->n_table == 13 && n_pos >= 0 && n_first >= 0 && n_second >= 0 && n_pos + n_first + n_second <= 64, TTUP_EFORMAT, "uplift blob: bad layer counts");
    net->max_batch = max_batch; net->max_len = max_len;
    Reader r{(const char*)blob + 40, blob_bytes - 40};
    int rc;
    const int D = net->D;
    {
        std::vector<float> v;
        TTUP_REQUIRE(r.take(&v, net->hd / 2), TTUP_EFORMAT, "uplift blob: bad inv_freq record");
        if ((rc = dev_copy(net.get(), v, &net->inv_freq_dev))) return rc;
    }
    if ((rc = make_vec(net.get(), r, D, &net->cls_dev))) return rc;
    if ((rc = make_mlp2(net.get(), r, 2, &net->ball_embed))) return rc;
    if ((rc = make_mlp2(net.get(), r, 2, &net->table_embed))) return rc;
    net->pos_layers.resize(n_pos); net->layers.resize(n_first); net->second.resize(n_second);
    for (auto& L : net->pos_layers) if ((rc = make_layer(net.get(), r, &L))) return rc;
    for (auto& L : net->layers) if ((rc = make_layer(net.get(), r, &L))) return rc;
    if ((rc = make_head(net.get(), r, &net->position_head))) return rc;
    for (auto& L : net->second) if ((rc = make_layer(net.get(), r, &L))) return rc;
    if ((rc = make_head(net.get(), r, &net->rotation_head))) return rc;
    TTUP_REQUIRE(r.left == 0, TTUP_EFORMAT, "uplift blob: %zu trailing bytes", r.left);
    make_stage(net.get(), net->pos_layers, &net->stage_pos);
    make_stage(net.get(), net->layers, &net->stage_first);
    make_stage(net.get(), net->second, &net->stage_second);
    {
        std::vector<float> tt(net->n_table);
        for (int n = 0; n < net->n_table; ++n) tt[n] = (float)n / 100.0f;       // arange(13) / (MAX_FPS/5), model.py:367
        if ((rc = dev_copy(net.get(), tt, &net->table_times_dev))) return rc;
        float* tr = nullptr;
        if ((rc = dev_alloc(net.get(), (size_t)net->n_table * net->hd, &tr))) return rc;
        net->table_rope = (float2*)tr;
        const long long n = (long long)net->n_table * (net->hd / 2);
        hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, net->table_times_dev, net->inv_freq_dev, net->table_rope, net->hd / 2, n);
        TTUP_LAUNCH_CHECK();
    }
    // scratch: chunk of trajectories such that the table stage holds at most ~2M tokens (7 GB of fp32 scratch at D=128)
    const long long per_traj = (long long)max_len * (net->n_table + 1);
    long long chunk = (2048 * 1024) / per_traj;
    if (chunk < 1) chunk = 1;
    if (chunk > max_batch) chunk = max_batch;
    net->chunk = (int)chunk;
    const size_t tokmax = (size_t)chunk * per_traj;
    const size_t bt = (size_t)chunk * (max_len + 1);
    if ((rc = dev_alloc(net.get(), tokmax * D, &net->x))) return rc;
    if ((rc = dev_alloc(net.get(), tokmax * 3 * D, &net->qkv))) return rc;
    if ((rc = dev_alloc(net.get(), tokmax * D, &net->att))) return rc;
    if ((rc = dev_alloc(net.get(), tokmax * D, &net->hid))) return rc;
    if ((rc = dev_alloc(net.get(), tokmax * D, &net->x2))) return rc;
    if ((rc = dev_alloc(net.get(), bt * D, &net->tok))) return rc;
    if ((rc = dev_alloc(net.get(), bt * D, &net->h1))) return rc;
    if ((rc = dev_alloc(net.get(), (size_t)chunk * net->n_table * D, &net->ttok))) return rc;
    if ((rc = dev_alloc(net.get(), bt, &net->m1))) return rc;
    if ((rc = dev_alloc(net.get(), bt, &net->m2))) return rc;
    if ((rc = dev_alloc(net.get(), (size_t)chunk * (net->n_table + 1), &net->tmask))) return rc;
    if ((rc = dev_alloc(net.get(), (size_t)chunk * net->n_table * 2, &net->txy))) return rc;
    float* fl = nullptr;
    if ((rc = dev_alloc(net.get(), bt * net->hd, &fl))) return rc;
    net->rope = (float2*)fl;
    if ((rc = dev_alloc(net.get(), 4, &fl))) return rc;
    net->flags_dev = (int*)fl;
    {
        // graph path: batches of up to GRAPH_TOKENS ball tokens (batch * len)
        const long long GRAPH_TOKENS = 1024;
        long long gt = (long long)max_batch * max_len < GRAPH_TOKENS ? (long long)max_batch * max_len : GRAPH_TOKENS;
        if (gt < max_len) gt = max_len;          // at least one trajectory of the longest length
        net->graph_tokens = gt;
        net->graphs_off = getenv("TTUP_UPLIFT_NO_GRAPH") != nullptr;
        const size_t nb = (size_t)(gt / 1 > max_batch ? max_batch : gt);          // trajectories a graph call can hold (len >= 1)
        if ((rc = dev_alloc(net.get(), (size_t)gt * 2, &net->g_ball))) return rc;
        if ((rc = dev_alloc(net.get(), nb * net->n_table * 3, &net->g_table))) return rc;
        if ((rc = dev_alloc(net.get(), (size_t)gt, &net->g_mask))) return rc;
        if ((rc = dev_alloc(net.get(), (size_t)gt, &net->g_times))) return rc;
        if ((rc = dev_alloc(net.get(), nb * 3, &net->g_rot))) return rc;
        if ((rc = dev_alloc(net.get(), (size_t)gt * 3, &net->g_pos))) return rc;
    }
    TTUP_HIP_CHECK(hipDeviceSynchronize());
    *out = net.release();
    return TTUP_OK;
}

extern "C" void ttup_uplift_destroy(ttup_uplift* net) {
    if (!net) return;
    (void)hipDeviceSynchronize();
    delete net;
}

extern "C" int ttup_uplift_forward(ttup_uplift* net, const float* ball_dev, const float* table_dev, const float* mask_dev,
                                   const float* times_dev, int batch, int len, float* rot_dev, float* pos_dev, int check_mask, void* stream) {
    TTUP_REQUIRE(net && ball_dev && table_dev && mask_dev && times_dev && rot_dev && pos_dev, TTUP_EINVAL, "ttup_uplift_forward: null pointer");
    TTUP_REQUIRE(batch >= 0 && batch <= net->max_batch, TTUP_EINVAL, "ttup_uplift_forward: batch %d outside [0,%d]", batch, net->max_batch);
    TTUP_REQUIRE(len > 0 && len <= net->max_len, TTUP_EINVAL, "ttup_uplift_forward: sequence length %d outside [1,%d]", len, net->max_len);
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0) return TTUP_OK;
    TTUP_HIP_CHECK(hipMemsetAsync(net->flags_dev, 0, sizeof(int), st));
    const long long cap = net->chunk;      // scratch is sized for `chunk` trajectories of max_len tokens
    bool done = false;
    if (!net->graphs_off && st != nullptr && batch <= cap && (long long)batch * len <= net->graph_tokens &&
        (net->graphs.size() < 32 || net->graphs.count({batch, len}))) {          // (at most 32 shapes are kept)
        ttup_uplift::GraphEntry& ge = net->graphs[{batch, len}];
        const size_t bt = (size_t)batch * len;
        auto copy_in = [&]() -> int {
            TTUP_HIP_CHECK(hipMemcpyAsync(net->g_ball, ball_dev, bt * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
            TTUP_HIP_CHECK(hipMemcpyAsync(net->g_table, table_dev, (size_t)batch * net->n_table * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
            TTUP_HIP_CHECK(hipMemcpyAsync(net->g_mask, mask_dev, bt * sizeof(float), hipMemcpyDeviceToDevice, st));
            TTUP_HIP_CHECK(hipMemcpyAsync(net->g_times, times_dev, bt * sizeof(float), hipMemcpyDeviceToDevice, st));
            return TTUP_OK;
        };
        if (!ge.exec && ge.seen >= 1) {
            // second call with this shape (the first one ran eagerly and set every kernel's attributes): capture
            if (int rc = copy_in()) return rc;
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int rc = forward_chunk(net, net->g_ball, net->g_table, net->g_mask, net->g_times, batch, len, net->g_rot, net->g_pos, st);
                ok = hipStreamEndCapture(st, &graph) == hipSuccess && rc == TTUP_OK && graph;
            }
            if (ok) ok = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) {
                if (getenv("TTUP_DEBUG")) fprintf(stderr, "ttup_uplift: graph capture failed (%s): eager from now on\n", hipGetErrorString(hipGetLastError()));
                (void)hipGetLastError(); ge.exec = nullptr; net->graphs_off = true;
            }          // this runtime cannot capture the forward: eager from now on
            else {
                TTUP_HIP_CHECK(hipGraphLaunch(ge.exec, st));
                done = true;
            }
        } else if (ge.exec) {
            if (int rc = copy_in()) return rc;
            TTUP_HIP_CHECK(hipGraphLaunch(ge.exec, st));
            done = true;
        }
        ge.seen++;
        if (done) {
            net->graph_replays++;
            TTUP_HIP_CHECK(hipMemcpyAsync(rot_dev, net->g_rot, (size_t)batch * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
            TTUP_HIP_CHECK(hipMemcpyAsync(pos_dev, net->g_pos, bt * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    for (int b0 = 0; b0 < batch && !done; b0 += (int)cap) {
        const int nb = batch - b0 < cap ? batch - b0 : (int)cap;
        const int rc = forward_chunk(net, ball_dev + (size_t)b0 * len * 2, table_dev + (size_t)b0 * net->n_table * 3, mask_dev + (size_t)b0 * len,
                                     times_dev + (size_t)b0 * len, nb, len, rot_dev + (size_t)b0 * 3, pos_dev + (size_t)b0 * len * 3, st);
        if (rc) return rc;
    }
    if (check_mask) {
        int flags = 0;
        TTUP_HIP_CHECK(hipMemcpyAsync(&flags, net->flags_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        TTUP_HIP_CHECK(hipStreamSynchronize(st));
        // reference: mask.min()==0 and mask.max()==1, else ValueError (model.py:541-546); the already-additive
        // {-1e9,0} format of the elif branch is not accepted here
        TTUP_REQUIRE(flags == 3, TTUP_EMASK, "wrong format for masks. Should be 0, 1 or -1e9, 0.");
    }
    return TTUP_OK;
}

// how the small-batch path is doing: out_host[0] = captured graphs, [1] = 1 when capturing failed on this runtime (eager from then
// on) or was switched off (TTUP_UPLIFT_NO_GRAPH), [2] = forwards served by a graph replay
extern "C" int ttup_uplift_graph_info(ttup_uplift* net, int* out_host3) {
    TTUP_REQUIRE(net && out_host3, TTUP_EINVAL, "ttup_uplift_graph_info: null pointer");
    int n = 0;
    for (auto& kv : net->graphs) n += kv.second.exec != nullptr;
    out_host3[0] = n; out_host3[1] = net->graphs_off ? 1 : 0; out_host3[2] = (int)net->graph_replays;
    return TTUP_OK;
}

// stage_x3_kernel launches issued (or captured into a graph) so far: all layers of a stage in one launch, small batches only
extern "C" int ttup_uplift_stage_info(ttup_uplift* net, long long* out_host) {
    TTUP_REQUIRE(net && out_host, TTUP_EINVAL, "ttup_uplift_stage_info: null pointer");
    *out_host = net->stage_launches;
    return TTUP_OK;
}

extern "C" int ttup_transform_rotationaxes(const float* rot_dev, const float* pos_dev, int batch, int len, float* out_dev, void* stream) {
    TTUP_REQUIRE(rot_dev && pos_dev && out_dev, TTUP_EINVAL, "ttup_transform_rotationaxes: null pointer");
    TTUP_REQUIRE(batch >= 0 && len >= 2, TTUP_EINVAL, "ttup_transform_rotationaxes: need at least two positions");
    if (batch == 0) return TTUP_OK;
    hipLaunchKernelGGL(rotationaxes_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, rot_dev, pos_dev, batch, len, out_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

#include "no_packed_fp32_end.h"
