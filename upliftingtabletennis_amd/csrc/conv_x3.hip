// fp32 convolution on the bf16 matrix pipe with SPLIT operands ("bf16 x 3"): every fp32 value v is split exactly into three bf16
// parts v = v1 + v2 + v3 (v1 = rne_bf16(v), v2 = rne_bf16(v - v1), v3 = rne_bf16(v - v1 - v2): 3 x 8 significant bits + the signs
// of the remainders cover the 24-bit significand), and a product w * x is evaluated as the six partial products
//     w1 x3 + w2 x2 + w3 x1 + w1 x2 + w2 x1 + w1 x1          (smallest first)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Each partial product of two bf16 values is exact in fp32; the three dropped
// ones (w2 x3, w3 x2, w3 x3) are below 2^-25 of |w x| -- under the rounding of a single fp32 fma (2^-24) -- so the result has the
// accuracy of an fp32 fma chain (measured against the fp32-MFMA kernel of csrc/conv_f32.hip and against the reference's torch-CPU
// heatmaps in tests/) at 2.7x its peak rate: the bf16 pipe does 16 x 16 x 32 MACs in 16 cycles, the fp32 pipe 16 x 16 x 4 in 32.
// This is the arithmetic of the parity path (TTUP_DTYPE_F32) and of the certified argmax's crops (csrc/certify.hip) since round 4;
// TTUP_F32_EXACT=1 selects the fp32-MFMA kernel instead (cross-check).
// Reference: the conv / BN(folded) / ReLU / residual call sites of balldetection/models/wasb.py:48-64, :85-105, :227-245, :446-451.
//
// Structure = the bf16 implicit-GEMM kernel of csrc/conv.hip (persistent workgroups walk (tile, channel chunk) items, the next item's
// global loads are issued before the MFMA loop of the current one): fp32 NHWC activations in, split while they are written to the
// LDS halo tile (three bf16 planes, pixel-major, the chunk swizzle of lds_off), weights split on the host and packed per MFMA
// fragment in three planes; fp32 NHWC out (+bias, +residual, ReLU).  Couts are processed in blocks of MT*16 (grid.y).
// The k order per output pixel (chunk, k-step, the MFMA's own order) does not depend on the tile or on the image size: a pixel
// computed on a crop equals the pixel computed on the whole frame bit for bit (what the certified argmax relies on).
#include "conv.h"
#include <stdlib.h>
#include <vector>

namespace ttup {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ConvX3Args {
    const float* src0; const float* src1; const bf16_t* wpack; const float* bias; const float* residual; float* dst;
    int c0, c1, nchunk0, nchunk, cout;
    int H, W, OH, OW, tiles_x, tiles_per_img, relu, batch;
    const int* n_active;
    // cone pruning (conv.h Roi): of a sample b with roi_flag[b] != 0 only the tiles [r_ty0, r_ty0 + r_nty) x [r_tx0, r_tx0 + r_ntx) are produced
    const int* roi_flag; int r_ty0, r_tx0, r_nty, r_ntx;
    int s_ty0, s_tx0, s_nty, s_ntx;          // the tile range of the samples with roi_flag[b] == 2 (conv.h Roi, class 2)
};

// The operand split is fp32 vector arithmetic (v - hi parts) that the compiler turns into `v_pk_add_f32 ... neg_lo neg_hi` -- packed
// fp32 with operand modifiers, the forms measured to return wrong values beside another kernel's LDS-fed MFMAs (csrc/common.h), and
// the crop passes run beside the bf16 CNN.  The subtractions are therefore issued as plain v_sub_f32 through inline asm (x3_sub);
// the rest of the unit keeps packed fp32 (element-wise forms without modifiers, measured safe: building the whole unit without
// packed fp32 cost 22 % of a crop pass).  tests/test_cabi.py scans the library's ISA for swizzled packed forms.
namespace {
__device__ __forceinline__ float x3_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// two fp32 -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned x3_pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// the same chunk swizzle as lds_off in csrc/conv.hip (CK = 32: 16-byte chunk index XOR bits 1..2 of the tile column)
template <int CK, int IW>
__device__ __forceinline__ int x3_off(int iy, int ix, int c8) {
    if (CK == 32) return ((iy * IW + ix) * 4 + (c8 ^ ((ix >> 1) & 3))) * 8;
    return ((iy * IW + ix) * (CK / 8) + c8) * 8;
}
// split 8 fp32 values into three packed-bf16 planes (v = p0 + p1 + p2 exactly; every subtraction is exact)
__device__ __forceinline__ void x3_split8(const f32x4& lo, const f32x4& hi, u32x4& p0, u32x4& p1, u32x4& p2) {
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned q0 = x3_pack2(a, b);
        const float ra = x3_sub(a, __uint_as_float(q0 << 16)), rb = x3_sub(b, __uint_as_float(q0 & 0xffff0000u));
        const unsigned q1 = x3_pack2(ra, rb);
        const float sa = x3_sub(ra, __uint_as_float(q1 << 16)), sb = x3_sub(rb, __uint_as_float(q1 & 0xffff0000u));
        p0[i] = q0; p1[i] = q1; p2[i] = x3_pack2(sa, sb);
    }
}

template <int CK, int MT, int KS, int S, int TH, int TW, int NW>
__global__ __launch_bounds__(NW * 64) void conv_x3_kernel(ConvX3Args a) {
    constexpr int NTHR = NW * 64;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int TAPS = KS * KS;
    constexpr int KSTEPS = (CK == 32) ? TAPS : (TAPS + 1) / 2;
    constexpr int NTW = TW / 16;
    constexpr int NT = TH * NTW / NW;         // N-tiles (16 pixels of one row) per wave
    constexpr int PAD = KS / 2;
    constexpr int IN_ELEMS = (IH * IW * CK + 7) & ~7;          // one plane
    constexpr int W_ELEMS = KSTEPS * MT * 64 * 8;              // one plane of one chunk
    constexpr int IN_UNITS = IH * IW * (CK / 8), IN_PT = (IN_UNITS + NTHR - 1) / NTHR;
    constexpr int W_UNITS = 3 * W_ELEMS / 8, W_PT = (W_UNITS + NTHR - 1) / NTHR;
    static_assert(TH * NTW % NW == 0, "tile must split over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_in = (bf16_t*)smem;                  // [3][IN_ELEMS]
    bf16_t* s_w = s_in + 3 * IN_ELEMS;             // [3][W_ELEMS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nchunk = a.nchunk;
    const int co_base = blockIdx.y * MT * 16;
    const bf16_t* wblk = a.wpack + (size_t)blockIdx.y * nchunk * 3 * W_ELEMS;
    int batch = a.batch;
    if (a.n_active) { const int na = *a.n_active; batch = na < batch ? na : batch; }
    const int tiles_x = a.tiles_x, tiles_per_img = a.tiles_per_img;
    constexpr int ty_off = 0, tx_off = 0;
    const int total_tiles = tiles_per_img * batch;
    const int my_tiles = total_tiles > (int)blockIdx.x ? (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int n_items = my_tiles * nchunk;
    // a tile of a pruned sample outside the op's region is skipped (wave-uniform: scalar loads of the sample's flag)
    auto skipped = [&](int item) {
        if (!a.roi_flag) return false;
        const int tl = blockIdx.x + (item / nchunk) * gridDim.x;
        const int b = tl / tiles_per_img, t = tl % tiles_per_img;
        const int f = a.roi_flag[b];
        if (f == 0) return false;
        const int ty = t / tiles_x, tx = t % tiles_x;
        if (f == 2) return ty < a.s_ty0 || ty >= a.s_ty0 + a.s_nty || tx < a.s_tx0 || tx >= a.s_tx0 + a.s_ntx;
        return ty < a.r_ty0 || ty >= a.r_ty0 + a.r_nty || tx < a.r_tx0 || tx >= a.r_tx0 + a.r_ntx;
    };
    auto next_item = [&](int item) {          // first item >= `item` that is not skipped (all chunks of a tile share the decision)
        while (item < n_items && item % nchunk == 0 && skipped(item)) item += nchunk;
        return item;
    };

    f32x4 pin[IN_PT][2];
    unsigned pin_ok = 0u;                              // bit k: unit k of the prefetched item lies inside the image (the others are written as zeros)
    u32x4 pw[W_PT];
    bool w_loaded = false, w_stored = false;          // single-chunk convs: the weights are staged with the first item the workgroup runs
    auto issue = [&](int item) {
        const int tl = blockIdx.x + (item / nchunk) * gridDim.x, chunk = item % nchunk;
        const int b = tl / tiles_per_img, t = tl % tiles_per_img;
        const int gy0 = (ty_off + t / tiles_x) * TH * S - PAD, gx0 = (tx_off + t % tiles_x) * TW * S - PAD;
        const bool first = chunk < a.nchunk0;
        const float* src = first ? a.src0 : a.src1;
        const int csrc = first ? a.c0 : a.c1;
        const int ch0 = (first ? chunk : chunk - a.nchunk0) * CK;
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * NTHR;
            const int c8 = u % (CK / 8), pix = u / (CK / 8);
            const int gy = gy0 + pix / IW, gx = gx0 + pix % IW;
            // branch-free: an invalid unit reads the tensor's first bytes and is zeroed (a branch around the load would serialise the loads)
            const bool ok = u < IN_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const float* p = ok ? src + ((size_t)(b * a.H + gy) * a.W + gx) * csrc + ch0 + c8 * 8 : a.src0;
            // the zeroing happens when the unit is committed to LDS: a select HERE makes the compiler wait for the load at once -- the
            // "prefetch" then sits out a whole memory round trip in front of every item's MFMA loop (round 5)
            pin[k][0] = *(const f32x4*)p; pin[k][1] = *(const f32x4*)(p + 4);
            pin_ok = ok ? pin_ok | (1u << k) : pin_ok & ~(1u << k);
        }
        if (nchunk > 1 || !w_loaded) {
            const u32x4* wsrc = (const u32x4*)(wblk + (size_t)chunk * 3 * W_ELEMS);
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * NTHR; pw[k] = wsrc[u < W_UNITS ? u : 0]; }
            w_loaded = true;
        }
    };
    auto commit = [&](int item) {
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * NTHR;
            if (u < IN_UNITS) {
                const int c8 = u % (CK / 8), pix = u / (CK / 8);
                u32x4 p0, p1, p2;
                const bool okk = (pin_ok >> k) & 1u;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                x3_split8(okk ? pin[k][0] : z, okk ? pin[k][1] : z, p0, p1, p2);
                bf16_t* d = s_in + x3_off<CK, IW>(pix / IW, pix % IW, c8);
                *(u32x4*)d = p0; *(u32x4*)(d + IN_ELEMS) = p1; *(u32x4*)(d + 2 * IN_ELEMS) = p2;
            }
        }
        if (nchunk > 1 || !w_stored) {
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * NTHR; if (u < W_UNITS) ((u32x4*)s_w)[u] = pw[k]; }
            w_stored = true;
        }
    };

    f32x4 bias[MT];          // seeds the accumulators
#pragma unroll
    for (int m = 0; m < MT; ++m) bias[m] = *(const f32x4*)(a.bias + co_base + g * 4 * MT + m * 4);

    // per-lane B-fragment bases: CK=32 -> one per tap column dx (k-step s = dy*KS+dx); CK=16 -> one per k-step (two taps)
    constexpr int NBB = (CK == 32) ? KS : KSTEPS;
    const bf16_t* bB[NBB];
#pragma unroll
    for (int k = 0; k < NBB; ++k) {
        int dy = 0, dx = k, c8 = g;
        if (CK != 32) {
            int tap = 2 * k + (g >> 1);
            if (tap > TAPS - 1) tap = TAPS - 1;     // padded k-group: weights are zero
            dy = tap / KS; dx = tap % KS; c8 = g & 1;
        }
        bB[k] = s_in + x3_off<CK, IW>(dy, n * S + dx, c8);
    }

    f32x4 acc[MT][NT];
    int item = next_item(0);
    if (item >= n_items) return;          // (workgroup-uniform)
    issue(item);
    // every path into the item loop has the prefetch registers complete; inside the loop they are waited for right behind the MFMA loop,
    // in front of the epilogue's stores (at the top of the next item, behind the stores, the wait would be an s_waitcnt vmcnt(0) that
    // drains them too: csrc/conv.hip prefetch_arrived)
    auto arrived = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) asm volatile("" :: "v"(pin[k][0]), "v"(pin[k][1]));
#pragma unroll
        for (int k = 0; k < W_PT; ++k) asm volatile("" :: "v"(pw[k]));
    };
    arrived();
    bool first = true;
    for (; item < n_items;) {
        const int nxt = next_item(item + 1);
        const int chunk = item % nchunk;
        if (!first) __syncthreads();            // every wave finished reading the previous item's LDS image
        first = false;
        commit(item);
        __syncthreads();
        if (nxt < n_items) issue(nxt);
        if (chunk == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = bias[m];
        }
        // the fragments of one k-step: 3 planes x (MT weight + NT pixel) 16-byte reads
        auto load_step = [&](int s, bf16x8 (&af)[3][MT], bf16x8 (&bfr)[3][NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m) af[p][m] = *(const bf16x8*)(s_w + p * W_ELEMS + ((s * MT + m) * 64 + lane) * 8);
            const bf16_t* bp = (CK == 32) ? bB[s % KS] : bB[s];
            const int dyc = (CK == 32) ? s / KS : 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nt = wave * NT + t;        // wave-uniform
                const int r = nt / NTW, cg = nt % NTW;
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[p][t] = *(const bf16x8*)(bp + p * IN_ELEMS + ((r * S + dyc) * IW + cg * 16 * S) * CK);
            }
        };
        // six partial products, smallest first; the MT x NT accumulators are independent chains between two dependent MFMAs
        auto mfma_step = [&](const bf16x8 (&af)[3][MT], const bf16x8 (&bfr)[3][NT]) __attribute__((always_inline)) {
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[q]][m], bfr[PB[q]][t], acc[m][t], 0, 0, 0);
        };
#ifdef TTUP_X3_FRAG_PIPELINE
        constexpr bool PIPE = MT == 1;          // two fragment sets are 24 (MT + NT) registers: only the 16-cout blocks have them to spare
#else
        // measured (round 5, full frame and 256 / 384-pixel crops): the pipelined form is 2-3 % SLOWER here (3.50 against 3.39 ms per
        // frame) -- it costs the 16-channel kernels an occupancy step (116 -> 132 registers: one workgroup per CU instead of two)
        constexpr bool PIPE = false;
#endif
        if constexpr (PIPE) {
            // pipelined like conv64_tile_mfma (csrc/conv.hip): the fragments of k-step s+1 are requested before the MFMAs of step s and a
            // scheduling barrier keeps the requests there -- same summation order per accumulator
            bf16x8 af[2][3][MT], bfr[2][3][NT];
            load_step(0, af[0], bfr[0]);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (s + 1 < KSTEPS) load_step(s + 1, af[(s + 1) & 1], bfr[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(af[s & 1], bfr[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                bf16x8 af[3][MT], bfr[3][NT];
                load_step(s, af, bfr);
                mfma_step(af, bfr);
            }
        }
        arrived();
        if (chunk != nchunk - 1) { item = nxt; continue; }
        // ---- epilogue: lane holds couts co_base + [g*4*MT, (g+1)*4*MT) of pixel n of each of its N-tiles
        const int tl = blockIdx.x + (item / nchunk) * gridDim.x;
        const int b = tl / tiles_per_img, tt = tl % tiles_per_img;
        const int oy0 = (ty_off + tt / tiles_x) * TH, ox0 = (tx_off + tt % tiles_x) * TW;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nt = wave * NT + t;
            const int oy = oy0 + nt / NTW, ox = ox0 + (nt % NTW) * 16 + n;
            if (oy >= a.OH || ox >= a.OW) continue;
            const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.cout + co_base + g * 4 * MT;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v = acc[m][t];
                if (a.residual) v += *(const f32x4*)(a.residual + o + m * 4);
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                *(f32x4*)(a.dst + o + m * 4) = v;
            }
        }
        item = nxt;
    }
}

template <int CK, int MT, int KS, int S, int TH, int TW, int NW>
int launch_x3(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int KSTEPS = (CK == 32) ? KS * KS : (KS * KS + 1) / 2;
    constexpr size_t SMEM = (size_t)(3 * ((IH * IW * CK + 7) & ~7) + 3 * KSTEPS * MT * 64 * 8) * 2;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    ConvX3Args a;
    a.src0 = (const float*)l.src0; a.src1 = (const float*)l.src1; a.wpack = (const bf16_t*)p.w3_dev; a.bias = p.bias_dev;
    a.residual = (const float*)l.residual; a.dst = (float*)l.dst;
    a.c0 = p.c0; a.c1 = p.cin_total - p.c0; a.nchunk0 = p.c0 / CK; a.nchunk = p.cin_total / CK; a.cout = p.cout;
    a.H = l.h; a.W = l.w; a.OH = (l.h + S - 1) / S; a.OW = (l.w + S - 1) / S;
    a.tiles_x = cdiv(a.OW, TW); a.tiles_per_img = a.tiles_x * cdiv(a.OH, TH);
    a.relu = l.relu; a.batch = l.batch; a.n_active = l.n_active;
    a.roi_flag = l.roi.flag; a.r_ty0 = a.r_tx0 = 0; a.r_nty = cdiv(a.OH, TH); a.r_ntx = a.tiles_x;
    if (l.roi.flag) {
        TTUP_REQUIRE(l.roi.y0 >= 0 && l.roi.x0 >= 0 && l.roi.y1 > l.roi.y0 && l.roi.x1 > l.roi.x0 && l.roi.y1 <= a.OH && l.roi.x1 <= a.OW, TTUP_EINVAL,
                     "conv x3: output region [%d,%d)x[%d,%d) outside %dx%d", l.roi.y0, l.roi.y1, l.roi.x0, l.roi.x1, a.OH, a.OW);
        a.r_ty0 = l.roi.y0 / TH; a.r_nty = cdiv(l.roi.y1, TH) - a.r_ty0;
        a.r_tx0 = l.roi.x0 / TW; a.r_ntx = cdiv(l.roi.x1, TW) - a.r_tx0;
    }
    a.s_ty0 = a.r_ty0; a.s_nty = a.r_nty; a.s_tx0 = a.r_tx0; a.s_ntx = a.r_ntx;
    if (l.roi.flag && l.roi.sy1 > 0) {
        TTUP_REQUIRE(l.roi.sy0 >= 0 && l.roi.sx0 >= 0 && l.roi.sy1 > l.roi.sy0 && l.roi.sx1 > l.roi.sx0 && l.roi.sy1 <= a.OH && l.roi.sx1 <= a.OW, TTUP_EINVAL,
                     "conv x3: class-2 output region [%d,%d)x[%d,%d) outside %dx%d", l.roi.sy0, l.roi.sy1, l.roi.sx0, l.roi.sx1, a.OH, a.OW);
        a.s_ty0 = l.roi.sy0 / TH; a.s_nty = cdiv(l.roi.sy1, TH) - a.s_ty0;
        a.s_tx0 = l.roi.sx0 / TW; a.s_ntx = cdiv(l.roi.sx1, TW) - a.s_tx0;
    }
    TTUP_REQUIRE(p.cout % (MT * 16) == 0 && p.mt3 == MT, TTUP_EINVAL, "conv x3: cout %d packed in blocks of %d, launched with %d", p.cout, p.mt3 * 16, MT * 16);
    const int blocks = p.cout / (MT * 16);
    const long long total = (long long)a.tiles_per_img * l.batch;
    if (total == 0) return TTUP_OK;
    int per_cu = (int)((160 * 1024) / SMEM);
    per_cu = per_cu > (NW == 8 ? 2 : 4) ? (NW == 8 ? 2 : 4) : (per_cu < 1 ? 1 : per_cu);
    int gx = 256 * per_cu / blocks;
    gx = gx < 1 ? 1 : gx;
    if (total < gx) gx = (int)total;
    if (int rc = ensure_max_lds((const void*)conv_x3_kernel<CK, MT, KS, S, TH, TW, NW>, SMEM)) return rc;
    hipLaunchKernelGGL((conv_x3_kernel<CK, MT, KS, S, TH, TW, NW>), dim3(gx, blocks), dim3(NW * 64), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

}  // namespace

// couts per block (MT * 16) of a conv of this shape: as many as the LDS budget of its tile allows
int conv_x3_block_mt(int cout, int k, int stride, int ck) {
    const int mt = cout / 16;
    if (k == 3 && stride == 2 && ck == 32) return mt > 2 ? 2 : mt;
    return mt > 4 ? 4 : mt;
}

bool conv_x3_supported(const PackedConv& p) {
    const int mt = p.cout / 16;
    return p.w3_dev && p.cout % 16 == 0 && (mt == 1 || mt == 2 || mt == 4 || mt == 8) && (p.ck == 16 || p.ck == 32) &&
           ((p.k == 3 && (p.stride == 1 || p.stride == 2)) || (p.k == 1 && p.stride == 1 && p.ck == 32));
}

int launch_conv_x3(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    const int mt = p.mt3;
#define X3_CASE(CK, MT, KS, S, TH, TW, NW) if (p.ck == CK && mt == MT) return launch_x3<CK, MT, KS, S, TH, TW, NW>(p, l, st)
    // pruned launches (the crop net): 16-pixel-wide tiles follow an op's region more closely than 8x32 ones (certification per varied
    // step 18.2 -> 16.7 ms on one box with the first four; TTUP_X3_WIDE=1 keeps the 32-wide tiles)
    static const bool wide = getenv("TTUP_X3_WIDE") != nullptr;
    if (!wide && l.roi.flag && p.k == 3 && p.stride == 1) {
        X3_CASE(32, 1, 3, 1, 16, 16, 8); X3_CASE(32, 2, 3, 1, 16, 16, 8); X3_CASE(32, 4, 3, 1, 8, 16, 8);
        X3_CASE(16, 1, 3, 1, 16, 16, 8); X3_CASE(16, 2, 3, 1, 16, 16, 8); X3_CASE(16, 4, 3, 1, 16, 16, 8);
    }
    if (!wide && l.roi.flag && p.k == 1 && p.stride == 1) {
        X3_CASE(32, 1, 1, 1, 16, 16, 8); X3_CASE(32, 2, 1, 1, 16, 16, 8); X3_CASE(32, 4, 1, 1, 16, 16, 8);
    }
    if (p.k == 3 && p.stride == 1) {
        X3_CASE(32, 1, 3, 1, 8, 32, 8); X3_CASE(32, 2, 3, 1, 8, 32, 8); X3_CASE(32, 4, 3, 1, 4, 32, 8);
        X3_CASE(16, 1, 3, 1, 8, 32, 8); X3_CASE(16, 2, 3, 1, 8, 32, 8); X3_CASE(16, 4, 3, 1, 8, 32, 8);
    } else if (p.k == 3 && p.stride == 2) {
        X3_CASE(32, 1, 3, 2, 4, 16, 4); X3_CASE(32, 2, 3, 2, 4, 16, 4);
        X3_CASE(16, 1, 3, 2, 4, 32, 8); X3_CASE(16, 2, 3, 2, 4, 32, 8); X3_CASE(16, 4, 3, 2, 4, 32, 8);
    } else if (p.k == 1 && p.stride == 1) {
        X3_CASE(32, 1, 1, 1, 8, 32, 8); X3_CASE(32, 2, 1, 1, 8, 32, 8); X3_CASE(32, 4, 1, 1, 8, 32, 8);
    }
#undef X3_CASE
    set_error("conv x3: k=%d stride=%d ck=%d cout block %d unsupported", p.k, p.stride, p.ck, mt * 16);
    return TTUP_EINVAL;
}

// host: split the folded fp32 weights into three bf16 planes in MFMA fragment order, per cout block:
// [block][chunk][plane][k-step][m][lane][8]; inside a block the cout permutation of csrc/conv.hip's packing (a lane ends with
// 4 * MT consecutive output channels of one pixel)
int pack_conv_x3(const std::vector<float>& w_tcc /* [tap][cin_total][cout] */, int cout, int cin_total, int c0, int k, int stride, PackedConv* out) {
    const int c1 = cin_total - c0, taps = k * k;
    out->w3_dev = nullptr; out->mt3 = 0;
    if (cout % 16 != 0 || c0 % 16 != 0 || c1 % 32 != 0) return TTUP_OK;          // not a shape of the matrix-pipe kernels: the exact kernels serve it
    const int ck = (c0 % 32 == 0) ? 32 : 16;
    if (ck == 16 && (c1 != 0 || k != 3)) return TTUP_OK;
    const int mt_all = cout / 16;
    if (!(mt_all == 1 || mt_all == 2 || mt_all == 4 || mt_all == 8)) return TTUP_OK;
    const int mt = conv_x3_block_mt(cout, k, stride, ck), blocks = mt_all / mt;
    const int ksteps = ck == 32 ? taps : (taps + 1) / 2, nchunk = cin_total / ck;
    const size_t plane = (size_t)ksteps * mt * 64 * 8;
    std::vector<bf16_t> w((size_t)blocks * nchunk * 3 * plane);
    for (int blk = 0; blk < blocks; ++blk)
        for (int c = 0; c < nchunk; ++c)
            for (int s = 0; s < ksteps; ++s)
                for (int m = 0; m < mt; ++m)
                    for (int l = 0; l < 64; ++l) {
                        const int i = l & 15, g = l >> 4;
                        const int co = blk * mt * 16 + (i >> 2) * (4 * mt) + m * 4 + (i & 3);
                        for (int j = 0; j < 8; ++j) {
                            int tap, ci;
                            if (ck == 32) { tap = s; ci = c * 32 + 8 * g + j; }
                            else { tap = 2 * s + (g >> 1); ci = c * 16 + 8 * (g & 1) + j; }
                            const float v = tap < taps ? w_tcc[((size_t)tap * cin_total + ci) * cout + co] : 0.f;
                            const bf16_t p0 = f32_to_bf16(v);
                            const float r1 = v - bf16_to_f32(p0);
                            const bf16_t p1 = f32_to_bf16(r1);
                            const float r2 = r1 - bf16_to_f32(p1);
                            const bf16_t p2 = f32_to_bf16(r2);
                            const size_t e = (((size_t)s * mt + m) * 64 + l) * 8 + j;
                            const size_t base = ((size_t)(blk * nchunk + c) * 3) * plane;
                            w[base + e] = p0; w[base + plane + e] = p1; w[base + 2 * plane + e] = p2;
                        }
                    }
    out->ck = ck; out->mt3 = mt;
    TTUP_HIP_CHECK(hipMalloc(&out->w3_dev, w.size() * sizeof(bf16_t)));
    TTUP_HIP_CHECK(hipMemcpy(out->w3_dev, w.data(), w.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    return TTUP_OK;
}

}  // namespace ttup
