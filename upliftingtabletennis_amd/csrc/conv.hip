// Implicit-GEMM convolution for the WASB/HRNet CNN on gfx950 (reference: balldetection/models/wasb.py
// conv/BN/ReLU/residual call sites :48-64, :85-105, :227-245, :446-451).
//
// GEMM view (per output tile):  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, cin)
//   A operand = weights  (M = cout, 16 per MFMA tile), pre-packed on the host in fragment order
//   B operand = pixels   (N = 16 consecutive output x of one row), read from an LDS halo tile
//   v_mfma_f32_16x16x32_bf16, fp32 accumulators; epilogue = +bias (+residual) (ReLU) -> bf16 NHWC.
// A lane ends up with 4*MT consecutive output channels of one pixel (the cout permutation is folded
// into the weight packing), so stores are 8..64 contiguous bytes per lane and a wave writes whole
// 16-pixel NHWC runs.
#include "conv.h"
#include <string.h>
#include <stdlib.h>

namespace ttup {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#ifdef TTUP_TIMING
// debug build only (tools/build_ablate.sh TIMING): phase timestamps (s_memtime) of wave 0.
//   TTUP_STAMP(k)        one-tile-per-workgroup kernels: slot k of workgroup blockIdx.x          (bb_chain2_kernel)
//   TTUP_STAMP_IT(id,it,k) persistent kernels: kernel id (0 stem, 1 bneck, 2 32-channel block), tile iteration it < 64 of workgroups < 32
__device__ unsigned long long ttup_tbuf[8192 * 8];
__device__ unsigned long long ttup_tbuf_it[3 * 32 * 64 * 8];
#define TTUP_BID ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)
#ifdef TTUP_TIMING_C16W
#define TTUP_STAMP(k) do { } while (0)          // (the per-wave stamps of csrc/chain16.h own the buffer)
#else
#define TTUP_STAMP(k) do { if (tid == 0 && TTUP_BID < 8192) ttup_tbuf[TTUP_BID * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifdef TTUP_TIMING_WAVES
// ... of EVERY wave of ONE kernel (id == TTUP_TIMING_WAVES): the buffer is read as [8 waves][12 workgroups][64 iterations][8 slots] (tools/wave_timing.py)
#define TTUP_STAMP_IT(id, it, k) do { if ((id) == TTUP_TIMING_WAVES && (tid & 63) == 0 && blockIdx.x < 12 && (it) < 64) ttup_tbuf_it[((((tid >> 6)) * 12 + blockIdx.x) * 64 + (it)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TTUP_STAMP_IT(id, it, k) do { if (tid == 0 && blockIdx.x < 32 && (it) < 64) ttup_tbuf_it[(((id) * 32 + blockIdx.x) * 64 + (it)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define TTUP_STAMP(k) do { } while (0)
#define TTUP_STAMP_IT(id, it, k) do { } while (0)
#endif

// MI355X: 8 XCDs with a private L2 each, workgroups dealt to them round-robin by linear id.  Persistent kernels walk tiles
// t = blockIdx.x + it * gridDim.x (gridDim.x a multiple of 8), so tile t runs on XCD t % 8 and raster neighbours -- which share
// halo rows / columns -- sit behind eight different L2s.  This remaps the sequence so that every XCD walks one contiguous
// eighth of the raster order: neighbours' halos become hits in the XCD's own L2 (PMC: 1.51 -> 1.40 GB of L2 fills per frame).
// Not applied in conv_mfma_kernel: its HBM-bound full-resolution conv gets 5-10 % slower with eight widely separated streams.
__device__ __forceinline__ int xcd_tile(int t, int total) {
#ifdef TTUP_NO_XCD_MAP
    return t;
#else
    const int main = total & ~7;
    return t < main ? (t & 7) * (main >> 3) + (t >> 3) : t;
#endif
}

struct ConvKArgs {
    const bf16_t* src0;
    const bf16_t* src1;
    const bf16_t* wpack;
    const float* bias;
    const bf16_t* residual;
    bf16_t* dst;
    int c0, c1;        // channels of the two sources
    int nchunk0, nchunk;  // chunks taken from src0, total chunks
    int H, W, OH, OW;
    int tiles_x, tiles_per_img, total_tiles;
    int relu;
    // fused 1x1 follower (F11): dst11 = relu(W11 . dst + b11), 64 -> 32 channels
    const bf16_t* w11; const float* bias11; bf16_t* dst11;
    // further fuse-layer terms added in the epilogue (wasb.py:236-243): res2 at the output resolution (the branch's own
    // tensor), res3 at 1/2^sh3 of it (a 1x1-conv'd lower branch, nearest-neighbour upsampled), both COUT channels
    const bf16_t* res2; const bf16_t* res3; int sh3;
    // conv64_kernel: linear 1x1 followers on the tile just produced (the fuse-layer convs 64 -> 16 / 64 -> 32 that feed the
    // higher-resolution branches, wasb.py:189-205: conv + BN, no ReLU)
    const bf16_t* wl16; const float* bl16; bf16_t* dl16;
    const bf16_t* wl32; const float* bl32; bf16_t* dl32;
    // conv_s2_pair_kernel: the second conv on the same input (16 -> 16), its own ReLU flag
    const bf16_t* wpack_b; const float* bias_b; bf16_t* dst_b; int relu_b;
    int xcd;           // conv_mfma_kernel: walk the tiles in the XCD-aware order of xcd_tile (stride-2 convs; see launch_mfma)
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two fp32 -> packed bf16 pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32)
// (as ONE vector conversion: two scalar casts come out as two conversions merged by a v_perm)
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// ReLU on a packed bf16 pair: bf16 is sign-magnitude, so as int16 every negative value (and -0) is < 0 (v_pk_max_i16)
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ unsigned relu_pk(unsigned p) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}

// LDS offset (in bf16 elements) of 8-channel group c8 of tile pixel (iy, ix).
// CK=32 (64 B per pixel): the 16-byte chunk index is XOR-swizzled with bits 1..2 of the tile column, which makes a
// 16-pixel ds_read_b128 conflict-free at every alignment (stride-1) and 2-way instead of 4-way at stride 2.
// CK=16 (32 B per pixel) is conflict-free as is.
template <int CK, int IW>
__device__ __forceinline__ int lds_off(int iy, int ix, int c8) {
    if (CK == 32) return ((iy * IW + ix) * 4 + (c8 ^ ((ix >> 1) & 3))) * 8;
    return ((iy * IW + ix) * (CK / 8) + c8) * 8;
}

// Copy UNITS 16-byte units from global memory into LDS with a 512-thread workgroup: every thread issues ALL of its loads before
// its first LDS store, i.e. one memory round trip for the block (a plain `for (u = tid; u < n; u += 512) dst[u] = src[u]` loop
// compiles to load / wait / store per iteration: UNITS / 512 serial round trips at the start of every persistent kernel).
template <int UNITS> struct StageRegs { u32x4 v[(UNITS + 511) / 512]; };
template <int UNITS>
__device__ __forceinline__ void stage_load_512(StageRegs<UNITS>& r, const bf16_t* src, int tid) {
#pragma unroll
    for (int k = 0; k < (UNITS + 511) / 512; ++k) { const int u = tid + k * 512; r.v[k] = u32x4{0u, 0u, 0u, 0u}; if (u < UNITS) r.v[k] = ((const u32x4*)src)[u]; }
}
template <int UNITS>
__device__ __forceinline__ void stage_store_512(bf16_t* dst, const StageRegs<UNITS>& r, int tid) {
#pragma unroll
    for (int k = 0; k < (UNITS + 511) / 512; ++k) { const int u = tid + k * 512; if (u < UNITS) ((u32x4*)dst)[u] = r.v[k]; }
}

// "These prefetched registers are needed HERE": an empty asm statement that takes them as inputs makes the compiler place its
// s_waitcnt for their loads at this point and treat them as complete afterwards.  The persistent kernels call it BEFORE an epilogue
// issues its stores: the vector-memory counter retires in order and the compiler cannot count stores that sit behind a branch, so a
// wait for prefetched loads that comes AFTER the stores is an s_waitcnt vmcnt(0) -- it drains the stores just issued, with every wave
// of the workgroup parked for a store round trip per tile (round 5: the stem spent 2.7 k of its 11.9 k cycles per tile there).
// A 32-bit per-lane offset the compiler must treat as unknown HERE: its zero-extension then happens next to the load that uses it, and
// "uniform 64-bit base + zext(32-bit lane offset)" is selected as ONE global_load with a scalar base (saddr) and a 32-bit vector
// offset.  Without it the extension is hoisted out of the tile loop (a register PAIR per offset) and every load gets a 64-bit add.
__device__ __forceinline__ unsigned opaque_u32(unsigned v) { asm volatile("" : "+v"(v)); return v; }

template <typename T, int N>
__device__ __forceinline__ void prefetch_arrived(const T (&r)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" :: "v"(r[k]));
}

// -DTTUP_PRIO_YOUNG (experiment, MI355X_MICROARCH.md "Static priority for the younger half"): waves 4-7 of an 8-wave workgroup lose
// the SIMD's issue arbitration to waves 0-3 (priority, then age); one s_setprio 1 for them at kernel start hands them the older half's timing
__device__ __forceinline__ void prio_young_half() {
#ifdef TTUP_PRIO_YOUNG
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
}

// Persistent, software-pipelined version: a workgroup walks work items (tile, channel chunk); the global loads of
// item i+1 (halo tile chunk + that chunk's weight fragments) are issued into registers BEFORE the MFMA loop of item i
// and written to LDS after it, so HBM/L2 latency hides behind the matrix work (single LDS buffer, two barriers per item).
// Single-chunk convs keep their weights resident in LDS across all tiles of the workgroup.
template <int CK, int COUT, int KS, int S, int TH, int TW, int NW, bool F11>
__global__ __launch_bounds__(NW * 64) void conv_mfma_kernel(ConvKArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int MT = COUT / 16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int TAPS = KS * KS;
    constexpr int KSTEPS = (CK == 32) ? TAPS : (TAPS + 1) / 2;
    constexpr int NTW = TW / 16;
    constexpr int NT = TH * NTW / NW;         // N-tiles per wave
    constexpr int PAD = KS / 2;
    constexpr int IN_ELEMS = IH * IW * CK;
    constexpr int W_ELEMS = KSTEPS * MT * 64 * 8;
    constexpr int IN_UNITS = IH * IW * (CK / 8), IN_PT = (IN_UNITS + NTHR - 1) / NTHR;
    constexpr int W_UNITS = W_ELEMS / 8, W_PT = (W_UNITS + NTHR - 1) / NTHR;
    static_assert(TH * NTW % NW == 0, "tile must split over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_in = (bf16_t*)smem;
    bf16_t* s_w = s_in + ((IN_ELEMS + 7) & ~7);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: row tests and row addresses on the scalar unit
    const int n = lane & 15, g = lane >> 4;
    const int nchunk = a.nchunk;
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_items = my_tiles * nchunk;
    const unsigned st_c = (unsigned)((n * COUT + g * 4 * MT) * 2);          // lane's byte offset inside a 16-pixel group of COUT-channel records

    u32x4 pin[IN_PT], pw[W_PT];
    // byte offsets of the thread's units from the tile's first halo pixel in src0 (see bb_chain_kernel).  Register budget: the 128-cout
    // variant sits at 252 of 256 with them and the stride-2 16 -> 64 conv at exactly 128 (two workgroups per CU; at 131 it was one
    // and 20 % slower) -- both only since the wave index is a scalar (readfirstlane) and the epilogue addresses take a scalar base
    constexpr bool FASTP = true;
    unsigned voff[FASTP ? IN_PT : 1];
    if constexpr (FASTP) {
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * NTHR;
            const int c8 = u % (CK / 8), pix = u / (CK / 8);
            voff[k] = u < IN_UNITS ? (unsigned)((((pix / IW) * a.W + pix % IW) * a.c0 + c8 * 8) * 2) : 0u;
        }
    }
    auto issue = [&](int item) {
        const int tl0 = blockIdx.x + (item / nchunk) * gridDim.x, chunk = item % nchunk;
        const int tl = a.xcd ? xcd_tile(tl0, a.total_tiles) : tl0;
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int gy0 = (t / a.tiles_x) * TH * S - PAD, gx0 = (t % a.tiles_x) * TW * S - PAD;
        const bool first = chunk < a.nchunk0;
        const bf16_t* src = first ? a.src0 : a.src1;
        const int csrc = first ? a.c0 : a.c1;
        const int ch0 = (first ? chunk : chunk - a.nchunk0) * CK;
#if !defined(TTUP_NO_FAST_PREFETCH) && !defined(TTUP_ABLATE_LOADS)
        if (FASTP && first && gy0 >= 0 && gy0 + IH <= a.H && gx0 >= 0 && gx0 + IW <= a.W) {          // halo tile inside the image: scalar base + lane constants
            const char* base = (const char*)(a.src0 + ((size_t)(b * a.H + gy0) * a.W + gx0) * a.c0 + ch0);
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) pin[k] = *(const u32x4*)(base + opaque_u32(voff[FASTP ? k : 0]));
        } else
#endif
        {
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) {
                const int u = tid + k * NTHR;
                const int c8 = u % (CK / 8), pix = u / (CK / 8);
                const int gy = gy0 + pix / IW, gx = gx0 + pix % IW;
                pin[k] = u32x4{0u, 0u, 0u, 0u};
#ifndef TTUP_ABLATE_LOADS
                if (u < IN_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    pin[k] = *(const u32x4*)(src + ((size_t)(b * a.H + gy) * a.W + gx) * csrc + ch0 + c8 * 8);
#endif
            }
        }
        if (nchunk > 1 || item == 0) {
            const u32x4* wsrc = (const u32x4*)(a.wpack + (size_t)chunk * W_ELEMS);
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * NTHR; if (u < W_UNITS) pw[k] = wsrc[u]; }
        }
    };
    auto commit = [&](int item) {
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * NTHR;
            if (u < IN_UNITS) { const int c8 = u % (CK / 8), pix = u / (CK / 8); *(u32x4*)(s_in + lds_off<CK, IW>(pix / IW, pix % IW, c8)) = pin[k]; }
        }
        if (nchunk > 1 || item == 0) {
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * NTHR; if (u < W_UNITS) ((u32x4*)s_w)[u] = pw[k]; }
        }
    };

    f32x4 bias[MT];          // seeds the accumulators
#pragma unroll
    for (int m = 0; m < MT; ++m) bias[m] = *(const f32x4*)(a.bias + g * 4 * MT + m * 4);

    // fused follower: its 4 weight fragments (2 k-steps x 2 m-tiles) stay in registers for the whole kernel
    bf16x8 af11[2][2];
    f32x4 bias11[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (F11) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 2; ++m) af11[k][m] = *(const bf16x8*)(a.w11 + ((k * 2 + m) * 64 + lane) * 8);
#pragma unroll
        for (int m = 0; m < 2; ++m) bias11[m] = *(const f32x4*)(a.bias11 + g * 8 + m * 4);
    }

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
        bB[k] = s_in + lds_off<CK, IW>(dy, n * S + dx, c8);
    }

    f32x4 acc[MT][NT];
    if (n_items <= 0) return;          // (workgroup-uniform)
    issue(0);
    prefetch_arrived(pin); prefetch_arrived(pw);          // every path into the loop has the prefetch registers complete (see prefetch_arrived)
    for (int item = 0; item < n_items; ++item) {
        const int chunk = item % nchunk;
        if (item > 0) __syncthreads();          // every wave finished reading the previous item's LDS image
        commit(item);
        __syncthreads();
        if (item + 1 < n_items) issue(item + 1);
        if (chunk == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = bias[m];
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            bf16x8 af[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = *(const bf16x8*)(s_w + ((s * MT + m) * 64 + lane) * 8);
            // lane-dependent part of the pixel-fragment address (tap column + channel chunk + swizzle) is precomputed in
            // bB[]; the N-tile / tap-row part below is a compile-time immediate
            const bf16_t* bp = (CK == 32) ? bB[s % KS] : bB[s];
            const int dyc = (CK == 32) ? s / KS : 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nt = wave * NT + t;        // wave-uniform
                const int r = nt / NTW, cg = nt % NTW;
                const bf16x8 bfr = *(const bf16x8*)(bp + ((r * S + dyc) * IW + cg * 16 * S) * CK);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#ifdef TTUP_ABLATE_MFMA
                    asm volatile("" :: "v"(af[m]), "v"(bfr));
#else
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][t], 0, 0, 0);
#endif
                }
            }
        }
#ifdef TTUP_ABLATE_EPILOGUE
        if (chunk != nchunk - 1 || a.H > 0) continue;
#endif
        // (no prefetch_arrived in front of the epilogue here: this kernel runs two to four workgroups per CU, another workgroup's MFMAs
        // cover a store drain at the top of the next item, and the HBM-bound 32 -> 32 conv at full resolution measured 4 % SLOWER with
        // the wait moved in front of its stores -- 0.204 against 0.196 ms, round 5)
        if (chunk != nchunk - 1) continue;
        // ---- epilogue: lane holds couts [g*4*MT, (g+1)*4*MT) of pixel n of each of its N-tiles
        const int tl0 = blockIdx.x + (item / nchunk) * gridDim.x;
        const int tl = a.xcd ? xcd_tile(tl0, a.total_tiles) : tl0;
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * TH, ox0 = (tt % a.tiles_x) * TW;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nt = wave * NT + t;
            const int oy = oy0 + nt / NTW, ox = ox0 + (nt % NTW) * 16 + n;
            if (oy >= a.OH || ox >= a.OW) continue;
            // element offset of the lane's first output channel: a wave-uniform part (scalar registers) + the lane constant -- the
            // loads and stores below then take a scalar base and a 32-bit lane offset instead of a 64-bit per-lane address chain
            const size_t ou = ((size_t)(b * a.OH + oy) * a.OW + ox0 + (nt % NTW) * 16) * COUT;
            const unsigned lc = opaque_u32(st_c);
            auto at = [&](const bf16_t* base) { return (bf16_t*)((char*)const_cast<bf16_t*>(base + ou) + lc); };
            float v[4 * MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m * 4 + r] = acc[m][t][r];
            // the terms are REQUESTED together and added in the reference's order (residual, res2, res3): a load issued behind the
            // previous term's wait costs one memory round trip per term
            // (wide outputs keep the one-term-at-a-time form: 3 x MT x 2 more registers do not fit beside 8 m-tiles of accumulators)
            constexpr int TM = MT <= 4 ? MT : 1;
            u32x2 tv[3][TM];
            auto load_term = [&](int k, const bf16_t* base) {
#pragma unroll
                for (int m = 0; m < TM; ++m) tv[k][m] = ((const u32x2*)base)[m];
            };
            auto add_term = [&](int k, const bf16_t* base) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const u32x2 rv = MT <= 4 ? tv[k][m < TM ? m : 0] : ((const u32x2*)base)[m];
                    v[m * 4 + 0] += bf16_to_f32((bf16_t)(rv.x & 0xffff));
                    v[m * 4 + 1] += bf16_to_f32((bf16_t)(rv.x >> 16));
                    v[m * 4 + 2] += bf16_to_f32((bf16_t)(rv.y & 0xffff));
                    v[m * 4 + 3] += bf16_to_f32((bf16_t)(rv.y >> 16));
                }
            };
            const bf16_t* t3 = a.res3 ? a.res3 + ((size_t)(b * (a.OH >> a.sh3) + (oy >> a.sh3)) * (a.OW >> a.sh3) + (ox >> a.sh3)) * COUT + g * 4 * MT : nullptr;
            if (MT <= 4) {
                if (a.residual) load_term(0, at(a.residual));
                if (a.res2) load_term(1, at(a.res2));
                if (a.res3) load_term(2, t3);
            }
            if (a.residual) add_term(0, at(a.residual));
            if (a.res2) add_term(1, at(a.res2));
            if (a.res3) add_term(2, t3);
            unsigned pk[2 * MT];
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) pk[i] = pack2(v[2 * i], v[2 * i + 1]);
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 2 * MT; ++i) pk[i] = relu_pk(pk[i]);
            }
            if (MT == 1) {
                *(u32x2*)at(a.dst) = u32x2{pk[0], pk[1]};
            } else {
#pragma unroll
                for (int q = 0; q < MT / 2; ++q) *(u32x4*)(at(a.dst) + q * 8) = u32x4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
            }
        }
        if (F11) {
            // ---- fused 1x1 follower on the tile just produced (Bottleneck conv1, wasb.py:88-90): the bf16 tile goes through
            // LDS (pixel-major, 128 B per pixel, chunks XOR-swizzled by the pixel index) and comes back as the B operand
            static_assert(!F11 || (COUT == 64 && TH * TW * 64 <= IN_ELEMS + W_ELEMS), "follower needs a 64-channel tile that fits the staging area");
            bf16_t* s_t = s_in;
            __syncthreads();                       // every wave is done with the staging area
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nt = wave * NT + t;
                const int p = (nt / NTW) * TW + (nt % NTW) * 16 + n;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x4 pk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned w = pack2(acc[2 * q + (i >> 1)][t][2 * (i & 1)], acc[2 * q + (i >> 1)][t][2 * (i & 1) + 1]);
                        pk[i] = a.relu ? relu_pk(w) : w;
                    }
                    *(u32x4*)(s_t + p * 64 + (((2 * g + q) ^ (p & 7)) << 3)) = pk;
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nt = wave * NT + t;
                const int p = (nt / NTW) * TW + (nt % NTW) * 16 + n;
                f32x4 c11[2] = {bias11[0], bias11[1]};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const bf16x8 bfr = *(const bf16x8*)(s_t + p * 64 + (((4 * k + g) ^ (p & 7)) << 3));
#pragma unroll
                    for (int m = 0; m < 2; ++m) c11[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af11[k][m], bfr, c11[m], 0, 0, 0);
                }
                const int oy = oy0 + nt / NTW, ox = ox0 + (nt % NTW) * 16 + n;
                if (oy >= a.OH || ox >= a.OW) continue;
                u32x4 pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) pk[i] = relu_pk(pack2(c11[i >> 1][2 * (i & 1)], c11[i >> 1][2 * (i & 1) + 1]));
                *(u32x4*)(a.dst11 + ((size_t)(b * a.OH + oy) * a.OW + ox) * 32 + g * 8) = pk;
            }
        }
    }
}

// ------------------------------------------------------------------ two stride-2 convs on one input
// Stage 3's fuse layer takes the full-resolution 16-channel branch down twice: 3x3 s2 16 -> 32 (the term of the half-resolution
// output, wasb.py:207-222 with i=1: conv + BN, the running fuse sum and ReLU in the epilogue) and 3x3 s2 16 -> 16 + ReLU (first
// conv of the chain towards the quarter resolution, i=2).  Both are HBM-bound on that 230-MB tensor (8 frames); here ONE
// workgroup pass stages the halo tile once and runs both: one read of the branch instead of two.  Same arithmetic per output
// as conv_mfma_kernel<16, COUT, 3, 2, 4, 32, 8> (same k-steps, same epilogue order).
__global__ __launch_bounds__(512) void conv_s2_pair_kernel(ConvKArgs a) {
    constexpr int CK = 16, KS = 3, S = 2, TH = 4, TW = 32, NW = 8, MTA = 2, MTB = 1;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int KSTEPS = 5, PAD = 1;
    constexpr int IN_ELEMS = IH * IW * CK;
    constexpr int IN_UNITS = IH * IW * (CK / 8), IN_PT = (IN_UNITS + 511) / 512;
    constexpr int WA_UNITS = KSTEPS * MTA * 64, WB_UNITS = KSTEPS * MTB * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_in = (bf16_t*)smem;
    bf16_t* s_wa = s_in + ((IN_ELEMS + 7) & ~7);
    bf16_t* s_wb = s_wa + WA_UNITS * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    StageRegs<WA_UNITS> wa; StageRegs<WB_UNITS> wb;
    stage_load_512<WA_UNITS>(wa, a.wpack, tid);
    stage_load_512<WB_UNITS>(wb, a.wpack_b, tid);
    u32x4 pin[IN_PT];
    unsigned pin_ok = 0u;
    auto issue = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int gy0 = (t / a.tiles_x) * TH * S - PAD, gx0 = (t % a.tiles_x) * TW * S - PAD;
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            const int c8 = u % (CK / 8), pix = u / (CK / 8);
            const int gy = gy0 + pix / IW, gx = gx0 + pix % IW;
            const bool ok = u < IN_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pin[k] = *(const u32x4*)(ok ? a.src0 + ((size_t)(b * a.H + gy) * a.W + gx) * CK + c8 * 8 : a.src0);      // branch-free: the loads go out together
            pin_ok = ok ? pin_ok | (1u << k) : pin_ok & ~(1u << k);       // zeroed when the unit is written to LDS: a select HERE would wait for the load at once (no prefetch)
        }
    };
    if (my_tiles <= 0) return;          // (workgroup-uniform)
    issue(0);
    stage_store_512<WA_UNITS>(s_wa, wa, tid);
    stage_store_512<WB_UNITS>(s_wb, wb, tid);
    f32x4 bias_a[MTA], bias_b;
#pragma unroll
    for (int m = 0; m < MTA; ++m) bias_a[m] = *(const f32x4*)(a.bias + g * 4 * MTA + m * 4);
    bias_b = *(const f32x4*)(a.bias_b + g * 4);
    // per-lane fragment bases, one per k-step (taps 2s | 2s+1 on lane groups 0-1 | 2-3; the tenth tap is a zero pad)
    const bf16_t* bB[KSTEPS];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
        int tap = 2 * k + (g >> 1);
        if (tap > 8) tap = 8;
        bB[k] = s_in + lds_off<CK, IW>(tap / KS, n * S + tap % KS, g & 1);
    }
    // the wave's 16-pixel group of the 4x32 tile: row wave / 2, column half wave % 2
    const int r = wave >> 1, cg = wave & 1;
    prefetch_arrived(pin);          // every path into the loop has the prefetch registers complete (see prefetch_arrived)
    for (int it = 0; it < my_tiles; ++it) {
        if (it > 0) __syncthreads();            // every wave finished reading the previous tile
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            const bool okk = (pin_ok >> k) & 1u;
            if (u < IN_UNITS) { const int c8 = u % (CK / 8), pix = u / (CK / 8); *(u32x4*)(s_in + lds_off<CK, IW>(pix / IW, pix % IW, c8)) = u32x4{okk ? pin[k].x : 0u, okk ? pin[k].y : 0u, okk ? pin[k].z : 0u, okk ? pin[k].w : 0u}; }
        }
        __syncthreads();
        if (it + 1 < my_tiles) issue(it + 1);
        f32x4 acc_a[MTA] = {bias_a[0], bias_a[1]}, acc_b = bias_b;
#pragma unroll
        for (int s5 = 0; s5 < KSTEPS; ++s5) {
            const bf16x8 bfr = *(const bf16x8*)(bB[s5] + ((r * S) * IW + cg * 16 * S) * CK);
#pragma unroll
            for (int m = 0; m < MTA; ++m) acc_a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(s_wa + ((s5 * MTA + m) * 64 + lane) * 8), bfr, acc_a[m], 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(s_wb + (s5 * 64 + lane) * 8), bfr, acc_b, 0, 0, 0);
        }
        prefetch_arrived(pin);          // the next tile's input is waited for in front of this tile's stores
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy = (tt / a.tiles_x) * TH + r, ox = (tt % a.tiles_x) * TW + cg * 16 + n;
        if (oy >= a.OH || ox >= a.OW) continue;
        const size_t opix = (size_t)(b * a.OH + oy) * a.OW + ox;
        {   // first conv: 32 outputs, lane holds couts g*8 .. g*8+7; fuse-layer terms in conv_mfma_kernel's order
            const size_t o = opix * 32 + g * 8;
            float v[8];
#pragma unroll
            for (int m = 0; m < MTA; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[m * 4 + q] = acc_a[m][q];
            auto add_term = [&](const bf16_t* base) {
                const u32x4 rv = *(const u32x4*)base;
                const unsigned w4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32((bf16_t)(w4[k] & 0xffff)); v[2 * k + 1] += bf16_to_f32((bf16_t)(w4[k] >> 16)); }
            };
            if (a.residual) add_term(a.residual + o);
            if (a.res2) add_term(a.res2 + o);
            if (a.res3) add_term(a.res3 + ((size_t)(b * (a.OH >> a.sh3) + (oy >> a.sh3)) * (a.OW >> a.sh3) + (ox >> a.sh3)) * 32 + g * 8);
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const unsigned w = pack2(v[2 * i], v[2 * i + 1]); pk[i] = a.relu ? relu_pk(w) : w; }
            *(u32x4*)(a.dst + o) = pk;
        }
        {   // second conv: 16 outputs, lane holds couts g*4 .. g*4+3
            const unsigned w0 = pack2(acc_b[0], acc_b[1]), w1 = pack2(acc_b[2], acc_b[3]);
            *(u32x2*)(a.dst_b + opix * 16 + g * 4) = u32x2{a.relu_b ? relu_pk(w0) : w0, a.relu_b ? relu_pk(w1) : w1};
        }
    }
}

// 3x3 64 -> 64 on an 8x32 tile from LDS (conv64_kernel and the stem's conv2): both 32-channel planes of the 10x34 halo tile and all
// 72 weight fragments are LDS-resident; a wave owns two vertically adjacent 16-pixel groups (rows 2q, 2q+1 of column half ch), whose
// four input rows are read once per (plane, tap column) and shared by both outputs.  18 k-steps (plane c, tap column dx, tap row dy --
// the summation order of every accumulator), 8 MFMAs each.
// PIPELINED (round 5): the fragments of step s+1 are requested BEFORE the MFMAs of step s, and a scheduling barrier keeps the
// requests where they are (the compiler otherwise sinks every ds_read to just in front of its first use: rrrr M wait M wait M ...,
// i.e. four reads covered by one MFMA, then the LDS latency in the open, 18 times per tile with only two waves per SIMD to hide it).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// hook(s): called in front of the MFMAs of k-step s (conv64_dma_kernel issues the next tile's DMA pieces there, under the matrix work)
template <int NPIX = 340, typename Hook = NoHook>          // NPIX: pixels per 32-channel plane of the halo tile (rows x 34): 340 for the 8-row tile, 204 for the stem's 4-row half tile
__device__ __forceinline__ void conv64_tile_mfma(f32x4 (&acc)[4][2], const bf16_t* const (&bB)[3], const bf16_t* s_w, int wave, int lane, Hook hook = Hook()) {
    constexpr int IW = 34;
    bf16x8 brow[2][4], af[2][4];
    auto load_b = [&](bf16x8 (&br)[4], int c, int dx) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) br[rr] = *(const bf16x8*)(bB[dx] + c * (NPIX * 32) + ((2 * (wave >> 1) + rr) * IW + (wave & 1) * 16) * 32);
    };
    auto load_a = [&](bf16x8 (&a4)[4], int st) __attribute__((always_inline)) {          // st = (c * 9 + dy * 3 + dx): the packed k-step
#pragma unroll
        for (int m = 0; m < 4; ++m) a4[m] = *(const bf16x8*)(s_w + ((st * 4 + m) * 64 + lane) * 8);
    };
    load_b(brow[0], 0, 0);
    load_a(af[0], 0);
#ifdef TTUP_ABL_MFMA32          // timing experiment (wrong results): the k-step's eight 16x16x32 MFMAs as four 32x32x16 ones on the same operand registers, two accumulator chains
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    f32x16 c32[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) c32[t][m * 4 + r] = acc[m][t][r];
#endif
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int dy = s % 3, gi = s / 3;                       // gi = (plane, tap column) group: c = gi / 3, dx = gi % 3
        if (s + 1 < 18) {
            const int s1 = s + 1, dy1 = s1 % 3, g1 = s1 / 3, c1 = g1 / 3, dx1 = g1 % 3;
            if (dy1 == 0) load_b(brow[g1 & 1], c1, dx1);
#ifdef TTUP_ABL_NOAFRAG          // timing experiment (wrong results): the weight fragments of step 0 serve every step -- what the LDS reads of the A operand cost
            af[s1 & 1][0] = af[s & 1][0]; af[s1 & 1][1] = af[s & 1][1]; af[s1 & 1][2] = af[s & 1][2]; af[s1 & 1][3] = af[s & 1][3];
#else
            load_a(af[s1 & 1], c1 * 9 + dy1 * 3 + dx1);
#endif
        }
        hook(s);
#ifndef TTUP_NO_FRAG_PIPELINE
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef TTUP_ABL_MFMA32
#pragma unroll
        for (int m = 0; m < 4; ++m) c32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][m], brow[gi & 1][dy + (m & 1)], c32[m & 1], 0, 0, 0);
#else
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s & 1][m], brow[gi & 1][dy + t], acc[m][t], 0, 0, 0);
#endif
#ifndef TTUP_NO_FRAG_PIPELINE
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifdef TTUP_ABL_MFMA32
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][t][r] = c32[t][m * 4 + r];
#endif
}

// ------------------------------------------------------------------ 3x3 64 -> 64 with resident weights
// The 64-channel BasicBlock convs of stages 3/4 (wasb.py:48-64) at 1/4 resolution: both 32-channel chunks of the weights
// (73.7 KB) stay in LDS for the life of the persistent workgroup, the whole 64-channel halo tile (10x34 px, 43.5 KB) is
// staged at once (register-prefetched one tile ahead), so a tile is 18 k-steps between two barriers and no weight byte
// moves inside the loop -- the generic kernel re-stages 36.8 KB of weights per (tile, chunk) item.
// L16 / L32: the 64 -> 16 / 64 -> 32 fuse-layer 1x1 convs ride in the epilogue (2 MFMAs per m-tile and pixel group on the bf16
// pairs just packed, follower K order permuted to the accumulator layout as in the stem): the branch tensor is not read again.
template <bool L16, bool L32>
__global__ __launch_bounds__(512) void conv64_kernel(ConvKArgs a) {
    constexpr int IH = 10, IW = 34, NPIX = IH * IW;
    constexpr int W_U = 2 * 9 * 4 * 64;                         // 16-byte units
    constexpr int IN_UNITS = NPIX * 8, IN_PT = (IN_UNITS + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_w = (bf16_t*)smem;                                // 73,728 B
    bf16_t* s_in = s_w + W_U * 8;                               // [2 chunks][340 px][32 ch]  43,520 B
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    StageRegs<W_U> wregs;
    stage_load_512<W_U>(wregs, a.wpack, tid);           // weights and the first tile travel together: one round trip before the loop
    f32x4 bias[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bias[m] = *(const f32x4*)(a.bias + g * 16 + m * 4);
    // follower fragments: a lane owns channels g*16 .. g*16+15 of its pixel, k-step k takes channels 16g + 8k + j from lane group
    // g; in the standard packing those sit at k-step g>>1, lane group 2(g&1)+k
    bf16x8 al16[2], al32[2][2];
    f32x4 bl16 = {0.f, 0.f, 0.f, 0.f}, bl32[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (L16) {
#pragma unroll
        for (int k = 0; k < 2; ++k) al16[k] = *(const bf16x8*)(a.wl16 + ((g >> 1) * 64 + n + 16 * ((g & 1) * 2 + k)) * 8);
        bl16 = *(const f32x4*)(a.bl16 + g * 4);
    }
    if (L32) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 2; ++m) al32[k][m] = *(const bf16x8*)(a.wl32 + (((g >> 1) * 2 + m) * 64 + n + 16 * ((g & 1) * 2 + k)) * 8);
#pragma unroll
        for (int m = 0; m < 2; ++m) bl32[m] = *(const f32x4*)(a.bl32 + g * 8 + m * 4);
    }
    const bf16_t* bB[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bB[dx] = s_in + lds_off<32, IW>(0, n + dx, g);
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    u32x4 pin[IN_PT];
    unsigned voff[IN_PT];          // byte offsets of the thread's units from the tile's first halo pixel (see bb_chain_kernel)
#pragma unroll
    for (int k = 0; k < IN_PT; ++k) {
        const int u = tid + k * 512;
        const int c8 = u & 7, pix = u >> 3;
        voff[k] = u < IN_UNITS ? (unsigned)((((pix / IW) * a.W + pix % IW) * 64 + c8 * 8) * 2) : 0u;
    }
    auto issue = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int gy0 = (t / a.tiles_x) * 8 - 1, gx0 = (t % a.tiles_x) * 32 - 1;
#ifndef TTUP_NO_FAST_PREFETCH
        if (gy0 >= 0 && gy0 + IH <= a.H && gx0 >= 0 && gx0 + IW <= a.W) {          // halo tile inside the image: scalar base + lane constants
            const char* base = (const char*)(a.src0 + ((size_t)(b * a.H + gy0) * a.W + gx0) * 64);
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) pin[k] = *(const u32x4*)(base + opaque_u32(voff[k]));
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            const int c8 = u & 7, pix = u >> 3;
            const int gy = gy0 + pix / IW, gx = gx0 + pix % IW;
            pin[k] = u32x4{0u, 0u, 0u, 0u};
            if (u < IN_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                pin[k] = *(const u32x4*)(a.src0 + ((size_t)(b * a.H + gy) * a.W + gx) * 64 + c8 * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            if (u < IN_UNITS) { const int c8 = u & 7, pix = u >> 3; *(u32x4*)(s_in + (c8 >> 2) * (NPIX * 32) + lds_off<32, IW>(pix / IW, pix % IW, c8 & 3)) = pin[k]; }
        }
    };
    if (my_tiles <= 0) return;          // (workgroup-uniform; the launcher never starts more workgroups than tiles)
    issue(0);
    stage_store_512<W_U>(s_w, wregs, tid);
    // every path into the tile loop has the prefetch registers (and every older load: biases, follower fragments) COMPLETE -- a path on
    // which one might be pending puts an s_waitcnt vmcnt(0) in front of its first use inside the loop, per tile (prefetch_arrived)
    prefetch_arrived(pin);
    for (int it = 0; it < my_tiles; ++it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * 8, ox0 = (tt % a.tiles_x) * 32;
        __syncthreads();                      // previous tile fully consumed (weights visible on the first pass)
        commit();
        __syncthreads();
        if (it + 1 < my_tiles) issue(it + 1);
        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc[m][0] = bias[m]; acc[m][1] = bias[m]; }
        conv64_tile_mfma(acc, bB, s_w, wave, lane);
        // the next tile's image (requested before the MFMA loop) is waited for HERE, in front of the epilogue's stores: at the loop top,
        // behind them, the same wait is an s_waitcnt vmcnt(0) that drains the stores as well (prefetch_arrived; unconditional).
        // (Round 5 also tried committing the next image here, behind a barrier, and requesting tile it+2 behind the epilogue: it kept the
        // vector-memory queue clean in the same way but ran 9 % slower -- 0.391 against 0.358 ms for the eight launches.)
        prefetch_arrived(pin);
        // the block input (residual) of BOTH pixel groups is requested before the first group's stores: a load behind a store would
        // make its wait drain that store too
        u32x4 rres[2][2];
        if (a.residual) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = oy0 + 2 * (wave >> 1) + t, ox = ox0 + (wave & 1) * 16 + n;
                const bool ok = oy < a.H && ox < a.W;
                const bf16_t* rp = a.residual + (ok ? ((size_t)(b * a.H + oy) * a.W + ox) * 64 + g * 16 : 0);      // branch-free: masked lanes read the tensor's first bytes
                rres[t][0] = *(const u32x4*)rp; rres[t][1] = *(const u32x4*)(rp + 8);
            }
        }
        // pass 1: both groups' outputs (bias + block input, ReLU, rounding) -- every residual value is consumed before the first store
        u32x4 pk[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[16];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m * 4 + r] = acc[m][t][r];
            if (a.residual) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x4 rv = rres[t][q];
                    const unsigned w4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[q * 8 + 2 * k] += bf16_to_f32((bf16_t)(w4[k] & 0xffff)); v[q * 8 + 2 * k + 1] += bf16_to_f32((bf16_t)(w4[k] >> 16)); }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const unsigned w = pack2(v[q * 8 + 2 * i], v[q * 8 + 2 * i + 1]); pk[t][q][i] = a.relu ? relu_pk(w) : w; }
        }
        // pass 2: stores and the fuse-layer followers
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int oy = oy0 + 2 * (wave >> 1) + t, ox = ox0 + (wave & 1) * 16 + n;
            const bool ok = oy < a.H && ox < a.W;
            if (!(L16 || L32) && !ok) continue;           // with followers every lane stays for the MFMAs; only the stores are masked
            const size_t opix = ok ? (size_t)(b * a.H + oy) * a.W + ox : 0;
            const size_t o = opix * 64 + g * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ok) *(u32x4*)(a.dst + o + q * 8) = pk[t][q];
            if (L16) {
                f32x4 c = bl16;
#pragma unroll
                for (int k = 0; k < 2; ++k) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al16[k], __builtin_bit_cast(bf16x8, pk[t][k]), c, 0, 0, 0);
                if (ok) *(u32x2*)(a.dl16 + opix * 16 + g * 4) = u32x2{pack2(c[0], c[1]), pack2(c[2], c[3])};
            }
            if (L32) {
                f32x4 c[2] = {bl32[0], bl32[1]};
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < 2; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al32[k][m], __builtin_bit_cast(bf16x8, pk[t][k]), c[m], 0, 0, 0);
                if (ok) *(u32x4*)(a.dl32 + opix * 32 + g * 8) = u32x4{pack2(c[0][0], c[0][1]), pack2(c[0][2], c[0][3]), pack2(c[1][0], c[1][1]), pack2(c[1][2], c[1][3])};
            }
        }
    }
}

template <bool L16, bool L32>
static int launch_conv64_t(const ConvKArgs& a, hipStream_t st) {
    constexpr size_t SMEM = (size_t)(2 * 9 * 4 * 64 * 8 + 2 * 340 * 32) * 2;
    if (int rc = ensure_max_lds((const void*)conv64_kernel<L16, L32>, SMEM)) return rc;
    const int grid = a.total_tiles < 256 ? a.total_tiles : 256;
    if (grid == 0) return TTUP_OK;
    kernel_note("conv64_kernel<%s, %s>", L16 ? "true" : "false", L32 ? "true" : "false");
    hipLaunchKernelGGL((conv64_kernel<L16, L32>), dim3(grid), dim3(512), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// ------------------------------------------------------------------ conv64 with LDS-DMA staging (round 5)
// The same conv as conv64_kernel (same k order, same epilogue: bit-identical results) with the halo tile staged by the DMA path
// (global_load_lds_dwordx4: memory -> LDS without passing through registers) into one of TWO tile buffers -- this kernel is the one
// persistent kernel with room for a second buffer (73.7 KB weights + 2 x 43.5 KB = 160,768 B of the 163,840).  Per tile: NO register
// staging (24 registers fewer), no commit pass, ONE barrier instead of two, the next tile in flight for a whole tile:
//     MFMA loop on buffer it & 1            (tile it+1 is landing in the other buffer)
//     wait for tile it+1 + barrier          -- in FRONT of this tile's stores (a wait behind them would drain them, prefetch_arrived)
//     request tile it+2 into buffer it & 1  (every wave is done reading it)
//     epilogue of tile it (stores)
// The DMA writes a wave's 64 x 16 bytes to CONSECUTIVE LDS addresses, so the swizzle of the tile image moves into the SOURCE address:
// the lane that fills slot j of pixel (iy, ix) fetches channel chunk j ^ ((ix >> 1) & 3).  Interior tiles: scalar tile base + six
// per-lane byte offsets computed once.  Border tiles (27 % at 1/4 resolution): the lane fetches the clamped pixel (always a valid
// address) and overwrites its slot with zeros once its own pieces have landed, before the barrier publishes the buffer.
template <bool L16, bool L32>
__global__ __launch_bounds__(512) void conv64_dma_kernel(ConvKArgs a) {
    prio_young_half();
    constexpr int IH = 10, IW = 34, NPIX = IH * IW;
    constexpr int W_U = 2 * 9 * 4 * 64;                         // 16-byte units
    constexpr int IN_UNITS = NPIX * 8, IN_PT = (IN_UNITS + 511) / 512;
    constexpr int BUF_ELEMS = 2 * NPIX * 32;                    // one tile image: [2 planes][340 px][32 ch]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_w = (bf16_t*)smem;                                // 73,728 B
    bf16_t* s_in = s_w + W_U * 8;                               // two tile images, 43,520 B each
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    StageRegs<W_U> wregs;
    stage_load_512<W_U>(wregs, a.wpack, tid);
    f32x4 bias[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bias[m] = *(const f32x4*)(a.bias + g * 16 + m * 4);
    // follower fragments: a lane owns channels g*16 .. g*16+15 of its pixel, k-step k takes channels 16g + 8k + j from lane group
    // g; in the standard packing those sit at k-step g>>1, lane group 2(g&1)+k
    bf16x8 al16[2], al32[2][2];
    f32x4 bl16 = {0.f, 0.f, 0.f, 0.f}, bl32[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (L16) {
#pragma unroll
        for (int k = 0; k < 2; ++k) al16[k] = *(const bf16x8*)(a.wl16 + ((g >> 1) * 64 + n + 16 * ((g & 1) * 2 + k)) * 8);
        bl16 = *(const f32x4*)(a.bl16 + g * 4);
    }
    if (L32) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 2; ++m) al32[k][m] = *(const bf16x8*)(a.wl32 + (((g >> 1) * 2 + m) * 64 + n + 16 * ((g & 1) * 2 + k)) * 8);
#pragma unroll
        for (int m = 0; m < 2; ++m) bl32[m] = *(const f32x4*)(a.bl32 + g * 8 + m * 4);
    }
    const bf16_t* bB[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bB[dx] = s_in + lds_off<32, IW>(0, n + dx, g);
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    // the thread's six units (16 bytes each) of a tile image: LDS slot u = k * 512 + tid, i.e. plane u / 1360, pixel (u % 1360) / 4, slot
    // j = u % 4 -- filled from channel chunk j ^ swizzle of that pixel.  voff: byte offset from the tile's first halo pixel (interior
    // tiles); pyxc: (iy << 16 | ix << 8 | first channel / 8) for the clamped addresses and the zero test of border tiles
    unsigned voff[IN_PT], pyxc[IN_PT];
#pragma unroll
    for (int k = 0; k < IN_PT; ++k) {
        const int u = tid + k * 512;
        const int uu = u < IN_UNITS ? u : 0;
        const int plane = uu / (NPIX * 4), r = uu % (NPIX * 4), pix = r >> 2, j = r & 3;
        const int iy = pix / IW, ix = pix % IW, c8 = plane * 4 + (j ^ ((ix >> 1) & 3));
        voff[k] = (unsigned)(((iy * a.W + ix) * 64 + c8 * 8) * 2);
        pyxc[k] = (unsigned)(iy << 16 | ix << 8 | c8);
    }
    unsigned zmask = 0;          // border tile in flight: bit k = the thread's unit k lies outside the image (zeroed once it has landed)
    // per-tile scalars of the tile being requested, then its pieces one at a time (piece k = units k * 512 .. + 511: one DMA per wave)
    int q_b = 0, q_gy0 = 0, q_gx0 = 0; bool q_in = false; const char* q_base = nullptr; char* q_dst = nullptr;
    auto issue_begin = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int t = tl % a.tiles_per_img;
        q_b = tl / a.tiles_per_img; q_gy0 = (t / a.tiles_x) * 8 - 1; q_gx0 = (t % a.tiles_x) * 32 - 1;
        q_in = q_gy0 >= 0 && q_gy0 + IH <= a.H && q_gx0 >= 0 && q_gx0 + IW <= a.W;
        q_base = (const char*)(a.src0 + ((size_t)(q_b * a.H + q_gy0) * a.W + q_gx0) * 64);
        q_dst = (char*)(s_in + (it & 1) * BUF_ELEMS);
        zmask = 0;
    };
    auto issue_piece = [&](int k) __attribute__((always_inline)) {
        if (k * 512 + wave * 64 >= IN_UNITS) return;          // (wave-uniform; the last piece is half a wave: lanes past the image stay out)
        auto* ldst = (__attribute__((address_space(3))) void*)(q_dst + (k * 512 + wave * 64) * 16);
        if (q_in) {
            if (k * 512 + tid < IN_UNITS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q_base + opaque_u32(voff[k])), ldst, 16, 0, 0);
        } else {
            const unsigned q = opaque_u32(pyxc[k]);
            const int gy = q_gy0 + (int)(q >> 16), gx = q_gx0 + (int)((q >> 8) & 255u);
            const int cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            if (cy != gy || cx != gx) zmask |= 1u << k;
            if (k * 512 + tid < IN_UNITS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.src0 + ((size_t)(q_b * a.H + cy) * a.W + cx) * 64 + (q & 255u) * 8), ldst, 16, 0, 0);
        }
    };
    // zero padding of a border tile: the thread overwrites ITS OWN out-of-image slots once its pieces have landed (the caller has waited
    // vmcnt(0): the DMA write of a slot and this write must not swap), before the barrier that publishes the buffer
    auto zero_fix = [&](int it) {
        if (__builtin_amdgcn_ballot_w64(zmask != 0) != 0) {          // (wave-uniform)
            char* dst = (char*)(s_in + (it & 1) * BUF_ELEMS);
#pragma unroll
            for (int k = 0; k < IN_PT; ++k)
                if ((zmask >> k & 1u) && k * 512 + tid < IN_UNITS) *(u32x4*)(dst + (k * 512 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    };
    if (my_tiles <= 0) return;          // (workgroup-uniform; the launcher never starts more workgroups than tiles)
    issue_begin(0);
#pragma unroll
    for (int k = 0; k < IN_PT; ++k) issue_piece(k);
    stage_store_512<W_U>(s_w, wregs, tid);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of tile 0 are in LDS
    zero_fix(0);
    // epilogue of one tile: bias + block input, ReLU, rounding, stores, fuse-layer followers (conv64_kernel's, verbatim)
    auto epilogue = [&](const f32x4 (&acc)[4][2], const u32x4 (&rres)[2][2], int b, int oy0, int ox0) __attribute__((always_inline)) {
        // pass 1: both groups' outputs (bias + block input, ReLU, rounding) -- every residual value is consumed before the first store
        u32x4 pk[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[16];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m * 4 + r] = acc[m][t][r];
            if (a.residual) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x4 rv = rres[t][q];
                    const unsigned w4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[q * 8 + 2 * k] += bf16_to_f32((bf16_t)(w4[k] & 0xffff)); v[q * 8 + 2 * k + 1] += bf16_to_f32((bf16_t)(w4[k] >> 16)); }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const unsigned w = pack2(v[q * 8 + 2 * i], v[q * 8 + 2 * i + 1]); pk[t][q][i] = a.relu ? relu_pk(w) : w; }
        }
        // pass 2: stores and the fuse-layer followers
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int oy = oy0 + 2 * (wave >> 1) + t, ox = ox0 + (wave & 1) * 16 + n;
            const bool ok = oy < a.H && ox < a.W;
            if (!(L16 || L32) && !ok) continue;           // with followers every lane stays for the MFMAs; only the stores are masked
            const size_t opix = ok ? (size_t)(b * a.H + oy) * a.W + ox : 0;
            const size_t o = opix * 64 + g * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ok) *(u32x4*)(a.dst + o + q * 8) = pk[t][q];
            if (L16) {
                f32x4 c = bl16;
#pragma unroll
                for (int k = 0; k < 2; ++k) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al16[k], __builtin_bit_cast(bf16x8, pk[t][k]), c, 0, 0, 0);
                if (ok) *(u32x2*)(a.dl16 + opix * 16 + g * 4) = u32x2{pack2(c[0], c[1]), pack2(c[2], c[3])};
            }
            if (L32) {
                f32x4 c[2] = {bl32[0], bl32[1]};
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < 2; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al32[k][m], __builtin_bit_cast(bf16x8, pk[t][k]), c[m], 0, 0, 0);
                if (ok) *(u32x4*)(a.dl32 + opix * 32 + g * 8) = u32x4{pack2(c[0][0], c[0][1]), pack2(c[0][2], c[0][3]), pack2(c[1][0], c[1][1]), pack2(c[1][2], c[1][3])};
            }
        }
    };
    // STAGGER (as in the stem; -DTTUP_CONV64_STAGGER): waves 4-7 -- the second wave of every SIMD -- run the epilogue of a tile at the
    // START of the next iteration, under the partner wave's MFMA loop, instead of beside the partner's own epilogue with the matrix pipe
    // idle; the accumulators and the block input stay in registers across the barrier.  MEASURED here (round 5): 0.340 against 0.3375 ms
    // for the eight launches -- no gain (the epilogue's vector work competes for the issue port the partner's MFMA loop needs): off
#ifdef TTUP_CONV64_STAGGER
    constexpr bool STAGGER = true;
#else
    constexpr bool STAGGER = false;
#endif
    const bool late = STAGGER && wave >= 4;
    f32x4 acc[4][2];
    u32x4 rres[2][2] = {};
    int eb = 0, eoy0 = 0, eox0 = 0;
    bool pending = false;
    for (int it = 0; it < my_tiles; ++it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * 8, ox0 = (tt % a.tiles_x) * 32;
        __syncthreads();                      // tile it visible (every wave waited for its own pieces); every wave is done with the other buffer
        if (late && pending) epilogue(acc, rres, eb, eoy0, eox0);
        // the block input (residual) of both pixel groups travels during the MFMA loop (the registers the staged tile no longer needs)
        if (a.residual) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = oy0 + 2 * (wave >> 1) + t, ox = ox0 + (wave & 1) * 16 + n;
                const bool ok = oy < a.H && ox < a.W;
                const bf16_t* rp = a.residual + (ok ? ((size_t)(b * a.H + oy) * a.W + ox) * 64 + g * 16 : 0);      // branch-free: masked lanes read the tensor's first bytes
                rres[t][0] = *(const u32x4*)rp; rres[t][1] = *(const u32x4*)(rp + 8);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc[m][0] = bias[m]; acc[m][1] = bias[m]; }
        const int sel = (it & 1) * BUF_ELEMS;
        const bf16_t* const bBc[3] = {bB[0] + sel, bB[1] + sel, bB[2] + sel};
        const bool more = it + 1 < my_tiles;
        if (more) issue_begin(it + 1);
        // the six pieces of tile it+1 go out UNDER this tile's matrix work, one every third k-step (an LDS-DMA instruction costs 60-180
        // cycles of issue, MI355X_MICROARCH.md: in one block in front of the epilogue they were 10 % of the kernel)
        // (placement measured: every third k-step from the first, second or third -- equal within noise; all six in the first six k-steps: 2 % slower)
        // (the hook's schedule -- a piece at k-steps 1, 4, ..., 16 of the 18 -- issues exactly six pieces; the waits below spell
        // s_waitcnt vmcnt(0) in gfx9 / gfx950 encoding: 0x0f70 = vmcnt 0 (bits 3:0 and 15:14), expcnt 7, lgkmcnt 15)
        static_assert(IN_PT == 6, "conv64_dma_kernel: the k-step hook issues pieces 0..5; a tile geometry with another piece count needs another schedule");
        conv64_tile_mfma<NPIX>(acc, bBc, s_w, wave, lane, [&](int s) __attribute__((always_inline)) { if (more && s % 3 == 1) issue_piece(s / 3); });
        // tile it+1 and the block input have landed -- waited for HERE, in front of this tile's stores (behind them the same wait drains them)
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        if (more) zero_fix(it + 1);
        if (late) { eb = b; eoy0 = oy0; eox0 = ox0; pending = true; }
        else epilogue(acc, rres, b, oy0, ox0);
    }
    if (late && pending) epilogue(acc, rres, eb, eoy0, eox0);
}

template <bool L16, bool L32>
static int launch_conv64_dma_t(const ConvKArgs& a, hipStream_t st) {
    constexpr size_t SMEM = (size_t)(2 * 9 * 4 * 64 * 8 + 2 * 2 * 340 * 32) * 2;
    if (int rc = ensure_max_lds((const void*)conv64_dma_kernel<L16, L32>, SMEM)) return rc;
    const int grid = a.total_tiles < 256 ? a.total_tiles : 256;
    if (grid == 0) return TTUP_OK;
    kernel_note("conv64_dma_kernel<%s, %s>", L16 ? "true" : "false", L32 ? "true" : "false");
    hipLaunchKernelGGL((conv64_dma_kernel<L16, L32>), dim3(grid), dim3(512), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

static int launch_conv64(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    ConvKArgs a;
    memset(&a, 0, sizeof a);
    a.src0 = (const bf16_t*)l.src0; a.wpack = (const bf16_t*)p.w_dev; a.bias = p.bias_dev; a.residual = (const bf16_t*)l.residual; a.dst = (bf16_t*)l.dst;
    a.H = l.h; a.W = l.w; a.OH = l.h; a.OW = l.w; a.relu = l.relu;
    a.tiles_x = cdiv(l.w, 32); a.tiles_per_img = a.tiles_x * cdiv(l.h, 8); a.total_tiles = a.tiles_per_img * l.batch;
    if (l.lin16) {
        TTUP_REQUIRE(l.lin16->cout == 16 && l.lin16->cin_total == 64 && l.lin16->k == 1 && l.lin16->ck == 32 && l.lin16_dst, TTUP_EINVAL, "conv64: bad 64->16 follower");
        a.wl16 = (const bf16_t*)l.lin16->w_dev; a.bl16 = l.lin16->bias_dev; a.dl16 = (bf16_t*)l.lin16_dst;
    }
    if (l.lin32) {
        TTUP_REQUIRE(l.lin32->cout == 32 && l.lin32->cin_total == 64 && l.lin32->k == 1 && l.lin32->ck == 32 && l.lin32_dst, TTUP_EINVAL, "conv64: bad 64->32 follower");
        a.wl32 = (const bf16_t*)l.lin32->w_dev; a.bl32 = l.lin32->bias_dev; a.dl32 = (bf16_t*)l.lin32_dst;
    }
    // default: the LDS-DMA form (conv64_dma_kernel), bit-identical to the register-staged conv64_kernel and 4-5 % faster (round 5, same
    // box, the eight launches of a micro-batch: 0.3405 against 0.3565 ms); TTUP_CONV64_DMA=0 selects the register-staged kernel
    static const bool dma = !(getenv("TTUP_CONV64_DMA") && getenv("TTUP_CONV64_DMA")[0] == '0');
    if (dma) {
        if (l.lin16 && l.lin32) return launch_conv64_dma_t<true, true>(a, st);
        if (l.lin16) return launch_conv64_dma_t<true, false>(a, st);
        if (l.lin32) return launch_conv64_dma_t<false, true>(a, st);
        return launch_conv64_dma_t<false, false>(a, st);
    }
    if (l.lin16 && l.lin32) return launch_conv64_t<true, true>(a, st);
    if (l.lin16) return launch_conv64_t<true, false>(a, st);
    if (l.lin32) return launch_conv64_t<false, true>(a, st);
    // (the 32x32x16-MFMA form of this conv, round 5: 0.8 % slower -- csrc/experiments/rejected_kernels.hip.inc)
    return launch_conv64_t<false, false>(a, st);
}

// ------------------------------------------------------------------ fused stem: conv1 + conv2 (+ Bottleneck conv1)
// Persistent workgroups (8 waves) keep ALL weights of the stem in LDS (conv1 20 KB + conv2 73.7 KB) and walk 8x32 tiles:
//   X0 halo tile (12x36 px, 16 ch, register-prefetched one tile ahead) -> conv1 3x3 9(16)->64 +ReLU on the 10x34 halo
//   region, kept in LDS as bf16 (never written to HBM) -> conv2 3x3 64->64 +ReLU straight from LDS (18 k-steps without a
//   barrier) -> T2 tile to HBM and, still in registers, into the 1x1 64->32 follower (Bottleneck conv1) -> A1 tile to HBM.
// Reference: wasb.py:446-451 (stem), :88-90 (Bottleneck conv1).  Intermediates are rounded to bf16 where the layer-wise
// path stores them, so results are bit-identical.
struct StemArgs {
    const bf16_t* x0;                 // (B,H,W,16), or in frames mode (NF > 0) the pre-processed frames (B+NF-1,H,W,4): sample b = frames b..b+NF-1
    const bf16_t* w1; const float* b1;      // conv1: CK=16 packing, 5 k-steps x 4 m-tiles
    const bf16_t* w2; const float* b2;      // conv2: CK=32 packing, 2 chunks x 9 k-steps x 4 m-tiles
    const bf16_t* w3; const float* b3;      // follower 1x1 64->32: 2 k-steps x 2 m-tiles
    bf16_t* t2; bf16_t* a1;
    int H, W, tiles_x, tiles_per_img, total_tiles;
};

// NF = 0: X0 comes as (B,H,W,16) records.  NF = 1 / 3 (frames mode): every frame is pre-processed ONCE into a 4-channel record
// (3 colours + 0) and a sample's X0 pixel is assembled in LDS from the NF frames it spans (slot f*4 + c; conv1's weights are
// packed in that channel order): the 16-channel per-triple tensor -- 3 copies of every frame plus 7 zero channels -- is never
// written or read (28.8 -> 7.2 MB of pre-processing output per frame, 28.8 -> 21.6 MB of stem input).
// K4 (NF = 3 only, round 5): conv1 in FOUR k-steps instead of five.  The X0 pixel record is the three frames' (B, G, R, 0) slots back to
// back -- 12 slots, 24 bytes -- so the three pixels under a tap row are 36 CONTIGUOUS slots of LDS: conv1's K dimension becomes
// 3 tap rows x 40 slots (36 + 4 that carry zero weights) = 120 -> 128 = 4 k-steps of 32, a fragment = 8 consecutive slots of one row
// (two 8-byte LDS reads: the records are 8-byte aligned).  The 16-slot records (each frame's 4 slots + 4 zero slots, two taps per
// k-step) need 5 k-steps for the 81 real products: 20 % of conv1's MFMAs and 4 KB of its weights gone.  Weights: StemArgs::w1 packed
// as a "1x1 conv with 128 inputs" in that slot order (csrc/wasb_net.hip).  Another fp32 summation order than the 5-step form (and
// than the layer-wise conv): results agree to bf16 rounding flips, like the other fused kernels (tests/test_gpu_parity.py).
template <int NF, bool K4 = false>
__global__ __launch_bounds__(512) void stem_kernel(StemArgs a) {
    prio_young_half();
    static_assert(!K4 || NF == 3, "the 4-step conv1 is the three-frame form");
    constexpr int XH = 12, XW = 36, TH1 = 10, TW1 = 34, NP1 = TH1 * TW1;       // X0 region, conv1 output region
    constexpr int KS1 = K4 ? 4 : 5;                                              // conv1 k-steps
    constexpr int XS = K4 ? 12 : 16;                                             // slots per X0 pixel record
    constexpr int W1_U = KS1 * 4 * 64, W2_U = 2 * 9 * 4 * 64;                    // 16-byte units
    constexpr int X_UNITS = NF ? XH * XW * NF : XH * XW * 2;                    // 8-byte (frames mode) or 16-byte units
    constexpr int X_PT = (X_UNITS + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_w1 = (bf16_t*)smem;                     // 20,480 B (16,384 B with K4)
    bf16_t* s_w2 = s_w1 + W1_U * 8;                   // 73,728 B
    bf16_t* s_t1 = s_w2 + W2_U * 8;                   // [2 chunks][340 px][32 ch]  43,520 B
    bf16_t* s_x = s_t1 + 2 * NP1 * 32;                // [432 px][16 slots] 13,824 B; K4: [432 px][12 slots] + 16 B of pad (the last fragment of the last pixel reads 4 slots past it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // (as a scalar -- readfirstlane -- the wave-dependent loops become branches: measured +3 ... 5 %)
    const int n = lane & 15, g = lane >> 4;
    StageRegs<W1_U> w1regs; StageRegs<W2_U> w2regs;
    stage_load_512<W1_U>(w1regs, a.w1, tid);            // stored to LDS after the first tile's loads have been issued (below)
    stage_load_512<W2_U>(w2regs, a.w2, tid);
    // Follower weights with the K order permuted to the conv2 accumulator layout: a lane owns channels g*16 .. g*16+15 of
    // its pixel, so k-step k takes channels 16g + 8k + j from lane group g -- the bf16 pairs it has just packed -- and the
    // T2 tile never goes through LDS.  In the standard packing those channels sit at k-step g>>1, lane group 2(g&1)+k.
    bf16x8 af3[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int m = 0; m < 2; ++m) af3[k][m] = *(const bf16x8*)(a.w3 + (((g >> 1) * 2 + m) * 64 + n + 16 * ((g & 1) * 2 + k)) * 8);
    // biases seed the accumulators (lane's channels g*16.. for the 64-channel convs, g*8.. for the follower)
    f32x4 b1[4], b2[4], b3[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) { b1[m] = *(const f32x4*)(a.b1 + g * 16 + m * 4); b2[m] = *(const f32x4*)(a.b2 + g * 16 + m * 4); }
#pragma unroll
    for (int m = 0; m < 2; ++m) b3[m] = *(const f32x4*)(a.b3 + g * 8 + m * 4);
    // conv1 per-lane tap offsets inside the X0 tile (CK=16: k-step s covers taps 2s and 2s+1)
    int koff1[KS1];
#pragma unroll
    for (int s5 = 0; s5 < KS1; ++s5) {
        if (K4) {          // k = 32 s + 8 g + j = 40 * (tap row) + slot: fragment (s, g) = slots o0 .. o0+7 of row r; k >= 120 carries zero weights (any valid address)
            const int kk0 = 32 * s5 + 8 * g, r = kk0 / 40, o0 = kk0 % 40;
            koff1[s5] = r < 3 ? r * XW * XS + o0 : 0;
        } else {
            int tap = 2 * s5 + (g >> 1); tap = tap > 8 ? 8 : tap; koff1[s5] = ((tap / 3) * XW + tap % 3) * 16 + (g & 1) * 8;
        }
    }
    // conv2 per-lane fragment bases inside one chunk plane of the T1 tile, one per tap column
    const bf16_t* bB[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bB[dx] = s_t1 + lds_off<32, TW1>(0, n + dx, g);

    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    u32x4 px[NF ? 1 : 2];
    u32x2 pf[NF ? X_PT : 1];
    if (NF) {         // slots no frame writes (the fourth record of a triple, three of four for a single frame; K4: the pad behind the tile) stay zero
        for (int u = tid; u < (K4 ? (XH * XW * XS * 2 + 16) / 16 : XH * XW * 2); u += 512) ((u32x4*)s_x)[u] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    // NF: byte offsets of the thread's (pixel, frame) records from the tile's first halo pixel in the triple's first frame (see bb_chain_kernel)
    unsigned xoff[NF ? X_PT : 1];
    if constexpr (NF != 0) {
#pragma unroll
        for (int k = 0; k < X_PT; ++k) {
            const int u = tid + k * 512;
            const int f = u % (NF ? NF : 1), pix = u / (NF ? NF : 1);
            xoff[k] = u < X_UNITS ? (unsigned)(((f * a.H + pix / XW) * a.W + pix % XW) * 8) : 0u;
        }
    }
    auto issue = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, t = tl % a.tiles_per_img;
        const int gy0 = (t / a.tiles_x) * 8 - 2, gx0 = (t % a.tiles_x) * 32 - 2;
        if (NF) {
#ifndef TTUP_NO_FAST_PREFETCH
            if (gy0 >= 0 && gy0 + XH <= a.H && gx0 >= 0 && gx0 + XW <= a.W) {          // halo tile inside the image: scalar base + lane constants
                const char* base = (const char*)(a.x0 + (((size_t)b * a.H + gy0) * a.W + gx0) * 4);
#pragma unroll
                for (int k = 0; k < X_PT; ++k) pf[k] = *(const u32x2*)(base + opaque_u32(xoff[k]));
                return;
            }
#endif
#pragma unroll
            for (int k = 0; k < X_PT; ++k) {
                const int u = tid + k * 512;
                const int f = u % (NF ? NF : 1), pix = u / (NF ? NF : 1);
                const int gy = gy0 + pix / XW, gx = gx0 + pix % XW;
                pf[k] = u32x2{0u, 0u};
                if (u < X_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    pf[k] = *(const u32x2*)(a.x0 + (((size_t)(b + f) * a.H + gy) * a.W + gx) * 4);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + k * 512;
            const int c8 = u & 1, pix = u >> 1;
            const int gy = gy0 + pix / XW, gx = gx0 + pix % XW;
            px[k & (NF ? 0 : 1)] = u32x4{0u, 0u, 0u, 0u};
            if (u < X_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                px[k & (NF ? 0 : 1)] = *(const u32x4*)(a.x0 + ((size_t)(b * a.H + gy) * a.W + gx) * 16 + c8 * 8);
        }
    };
    // The X0 tile of tile it+1 is committed to LDS in the MIDDLE of iteration it: behind the barrier that ends conv1 (the last
    // reader of the X0 buffer) and BEFORE conv2's epilogue issues its stores, and the loads of tile it+2 are requested right there.
    // Committed at the loop top -- behind the epilogue -- the wait for the prefetched loads was an s_waitcnt vmcnt(0) that also
    // drained the T2 / A1 stores just issued (the counter retires in order, and the compiler cannot count stores that sit behind
    // a branch): 2.7 k of the tile's 11.9 k cycles with every wave of the CU parked (round 5).
    auto commit = [&]() {
        if (NF) {
#pragma unroll
            for (int k = 0; k < X_PT; ++k) {
                const int u = tid + k * 512;
                if (u < X_UNITS) *(u32x2*)(s_x + (u / (NF ? NF : 1)) * XS + (u % (NF ? NF : 1)) * 4) = pf[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int u = tid + k * 512; if (u < X_UNITS) ((u32x4*)s_x)[u] = px[k & (NF ? 0 : 1)]; }
        }
    };
    // T2 tile to HBM; follower A1 = relu(W3 . T2 + b3), 64 -> 32, straight from the packed registers
    auto epilogue = [&](const f32x4 (&acc)[4][2], int b, int oy0, int ox0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = 2 * (wave >> 1) + t, cg = wave & 1;
            const int oy = oy0 + r, ox = ox0 + cg * 16 + n;
            const bool ok = oy < a.H && ox < a.W;
            u32x4 pk[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pk[q][i] = relu_pk(pack2(acc[2 * q + (i >> 1)][t][2 * (i & 1)], acc[2 * q + (i >> 1)][t][2 * (i & 1) + 1]));
#ifndef TTUP_ABLATE_SG
                if (ok) *(u32x4*)(a.t2 + ((size_t)(b * a.H + oy) * a.W + ox) * 64 + g * 16 + q * 8) = pk[q];
#endif
            }
#ifndef TTUP_ABLATE_S3
            f32x4 c3[2] = {b3[0], b3[1]};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m) c3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af3[k][m], __builtin_bit_cast(bf16x8, pk[k]), c3[m], 0, 0, 0);
            if (ok) {
                u32x4 po;
#pragma unroll
                for (int i = 0; i < 4; ++i) po[i] = relu_pk(pack2(c3[i >> 1][2 * (i & 1)], c3[i >> 1][2 * (i & 1) + 1]));
                *(u32x4*)(a.a1 + ((size_t)(b * a.H + oy) * a.W + ox) * 32 + g * 8) = po;
            }
#endif
        }
    };
#ifdef TTUP_NO_STAGGER
    constexpr bool STAGGER = false;
#else
    constexpr bool STAGGER = K4;
#endif
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
    f32x4 acc[4][2];
    int eb = 0, eoy0 = 0, eox0 = 0;
    bool pending = false;
    if (my_tiles <= 0) return;          // (workgroup-uniform; the launcher never starts more workgroups than tiles)
    issue(0);
    stage_store_512<W1_U>(s_w1, w1regs, tid);
    stage_store_512<W2_U>(s_w2, w2regs, tid);
    commit();                           // unconditional: its wait retires every older load (biases, follower fragments) on every path into the loop
    if (my_tiles > 1) issue(1);
    for (int it = 0; it < my_tiles; ++it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * 8, ox0 = (tt % a.tiles_x) * 32;
        const bool t1_inside = oy0 >= 1 && oy0 + 9 <= a.H && ox0 >= 1 && ox0 + 33 <= a.W;          // the whole 10x34 conv1 region lies inside the image
        TTUP_STAMP_IT(0, it, 0);
        TTUP_STAMP_IT(0, it, 1);
        // ONE barrier covers "X0 tile complete" (committed in the middle of the previous iteration) and "previous conv2 done reading
        // the T1 tile" (and the weights on the first pass)
        __syncthreads();
        TTUP_STAMP_IT(0, it, 2);
        // ---------------- conv1 on the 10x34 region (22 groups of 16 pixels, linear pixel index)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int j = wave + 8 * t;
            if (j >= 22) continue;
            const int p = j * 16 + n, pc = p < NP1 ? p : NP1 - 1;
            const int y = pc / TW1, x = pc % TW1;
            const bf16_t* xb = s_x + (y * XW + x) * XS;
            f32x4 acc[4] = {b1[0], b1[1], b1[2], b1[3]};
#ifndef TTUP_ABLATE_S1
#pragma unroll
            for (int s5 = 0; s5 < KS1; ++s5) {
                bf16x8 bfr;
                if (K4) {          // 8-byte aligned: two ds_read_b64
                    const u32x2 lo = *(const u32x2*)(xb + koff1[s5]), hi = *(const u32x2*)(xb + koff1[s5] + 4);
                    bfr = __builtin_bit_cast(bf16x8, u32x4{lo.x, lo.y, hi.x, hi.y});
                } else bfr = *(const bf16x8*)(xb + koff1[s5]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const bf16x8 af = *(const bf16x8*)(s_w1 + ((s5 * 4 + m) * 64 + lane) * 8);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[m], 0, 0, 0);
                }
            }
#endif
            if (p < NP1) {
                // conv2's zero padding: conv1 outputs outside the image are zeros.  Only border tiles have any (wave-uniform test on the
                // scalar unit): interior tiles skip the per-lane position test and the eight selects per pixel group (round 5: the
                // vector issue port is what these kernels run out of)
                const int gy = oy0 - 1 + y, gx = ox0 - 1 + x;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x4 pk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk[i] = relu_pk(pack2(acc[2 * q + (i >> 1)][2 * (i & 1)], acc[2 * q + (i >> 1)][2 * (i & 1) + 1]));
                    if (!t1_inside) {
                        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk[i] = inside ? pk[i] : 0u;
                    }
                    // lane's channels g*16 + q*8 .. +7  ->  chunk plane (g>>1), 16-byte chunk (g&1)*2+q
                    *(u32x4*)(s_t1 + (g >> 1) * (NP1 * 32) + lds_off<32, TW1>(y, x, (g & 1) * 2 + q)) = pk;
                }
            }
        }
        TTUP_STAMP_IT(0, it, 3);
        __syncthreads();
        commit();                                        // conv1 was the X0 buffer's last reader; unconditional (see conv64_kernel): on the last tile a stale image nobody reads
        if (it + 2 < my_tiles) issue(it + 2);
        TTUP_STAMP_IT(0, it, 4);
        // ---------------- conv2 on the 8x32 tile, both 32-channel planes straight from LDS
        // STAGGER (waves 4-7, the second wave of every SIMD): the epilogue of a tile is deferred to the start of the NEXT tile's conv2
        // phase, so it runs under the partner wave's MFMA loop instead of beside the partner's own epilogue (both waves of a SIMD
        // otherwise leave the matrix pipe idle together); the accumulators stay in registers across the tile boundary
        if (STAGGER && late && pending) epilogue(acc, eb, eoy0, eox0);
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc[m][0] = b2[m]; acc[m][1] = b2[m]; }
#ifndef TTUP_ABLATE_S2
        static_assert(TW1 == 34 && NP1 == 340, "conv64_tile_mfma's tile");
        conv64_tile_mfma(acc, bB, s_w2, wave, lane);
#endif
        TTUP_STAMP_IT(0, it, 5);
        if (STAGGER && late) { eb = b; eoy0 = oy0; eox0 = ox0; pending = true; }
        else epilogue(acc, b, oy0, ox0);
    }
    if (STAGGER && late && pending) epilogue(acc, eb, eoy0, eox0);
}

int launch_stem(const PackedConv& p1, const PackedConv& p2, const PackedConv& p3, const void* x0, void* t2, void* a1,
                int batch, int h, int w, hipStream_t st, int frames_per_sample) {
    TTUP_REQUIRE((p1.cout == 64 && p1.cin_total == 16 && p1.k == 3 && p1.stride == 1 && p1.ck == 16) ||
                 (p1.cout == 64 && p1.cin_total == 128 && p1.k == 1 && p1.ck == 32), TTUP_EINVAL, "stem: unexpected conv1 shape");
    TTUP_REQUIRE(p2.cout == 64 && p2.cin_total == 64 && p2.k == 3 && p2.stride == 1 && p2.ck == 32, TTUP_EINVAL, "stem: unexpected conv2 shape");
    TTUP_REQUIRE(p3.cout == 32 && p3.cin_total == 64 && p3.k == 1 && p3.ck == 32, TTUP_EINVAL, "stem: unexpected follower shape");
    StemArgs a;
    a.x0 = (const bf16_t*)x0; a.w1 = (const bf16_t*)p1.w_dev; a.b1 = p1.bias_dev; a.w2 = (const bf16_t*)p2.w_dev; a.b2 = p2.bias_dev;
    a.w3 = (const bf16_t*)p3.w_dev; a.b3 = p3.bias_dev; a.t2 = (bf16_t*)t2; a.a1 = (bf16_t*)a1;
    a.H = h; a.W = w; a.tiles_x = cdiv(w, 32); a.tiles_per_img = a.tiles_x * cdiv(h, 8); a.total_tiles = a.tiles_per_img * batch;
    constexpr size_t SMEM = (size_t)(5 * 4 * 64 * 8 + 2 * 9 * 4 * 64 * 8 + 2 * 340 * 32 + 432 * 16) * 2;          // (the 4-step form needs 7.5 KB less; one size for all)
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    TTUP_REQUIRE(frames_per_sample == 0 || frames_per_sample == 1 || frames_per_sample == 3, TTUP_EINVAL, "stem: frames per sample must be 0 (X0 records), 1 or 3");
    const bool k4 = p1.k == 1;          // conv1 packed as 128 slots x 1 tap: the 4-step three-frame form (csrc/wasb_net.hip)
    TTUP_REQUIRE(!k4 || frames_per_sample == 3, TTUP_EINVAL, "stem: the 4-step conv1 packing is the three-frame form");
    const void* kfn = k4 ? (const void*)stem_kernel<3, true> : frames_per_sample == 3 ? (const void*)stem_kernel<3> : frames_per_sample == 1 ? (const void*)stem_kernel<1> : (const void*)stem_kernel<0>;
    if (int rc = ensure_max_lds(kfn, SMEM)) return rc;
    const int grid = a.total_tiles < 256 ? a.total_tiles : 256;
    if (grid == 0) return TTUP_OK;
    // (a two-wave-group pipeline of the stem, round 5: 14 % slower -- csrc/experiments/rejected_kernels.hip.inc)
    kernel_note(k4 ? "stem_kernel<3, true>" : frames_per_sample == 3 ? "stem_kernel<3, false>" : frames_per_sample == 1 ? "stem_kernel<1, false>" : "stem_kernel<0, false>");
    if (k4) hipLaunchKernelGGL((stem_kernel<3, true>), dim3(grid), dim3(512), SMEM, st, a);
    else if (frames_per_sample == 3) hipLaunchKernelGGL(stem_kernel<3>, dim3(grid), dim3(512), SMEM, st, a);
    else if (frames_per_sample == 1) hipLaunchKernelGGL(stem_kernel<1>, dim3(grid), dim3(512), SMEM, st, a);
    else hipLaunchKernelGGL(stem_kernel<0>, dim3(grid), dim3(512), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// ------------------------------------------------------------------ fused Bottleneck tail + transition1
// One workgroup (8 waves) produces an 8x32 tile of transition1[0] (3x3 s1 128->16) and the matching 4x16 tile of
// transition1[1] (3x3 s2 128->32) without the 128-channel layer1 tensor ever leaving the CU:
//   phase 1  layer1 = relu(conv3(A2) + downsample(T2) + b) on the 10x34 halo tile (1x1, K = 32+64, 128 couts),
//            rounded to bf16 into LDS exactly as the unfused path rounds it into HBM;
//   phase 2a 3x3 s1 over the LDS tile -> B0;   phase 2b 3x3 s2 over the same tile -> B1.
// Reference: wasb.py:96-105 (conv3/bn3 + downsample + add + relu), :454-459 (transition1).
struct FusedArgs {
    const bf16_t* a2; const bf16_t* t2;           // (B,H,W,32), (B,H,W,64)
    const bf16_t* w1; const float* b1;            // two-source 1x1 -> 128 (3 chunks)
    const bf16_t* w5; const float* b5;            // 3x3 s1 128 -> 16 (4 chunks x 9 steps)
    const bf16_t* w6; const float* b6;            // 3x3 s2 128 -> 32 (4 chunks x 9 steps x 2 m-tiles)
    bf16_t* b0; bf16_t* b1o;
    int H, W, tiles_x, tiles_per_img, total_tiles;
};

__device__ __forceinline__ int l1_off(int pix, int c8) { return pix * 128 + ((c8 ^ (pix & 15)) << 3); }
__device__ __forceinline__ int st_off(int pix, int c8) { return pix * 32 + ((c8 ^ ((4 - ((pix >> 2) & 3)) & 3)) << 3); }

// No weight traffic inside the tile loop: W1 and W5 stay in LDS for the life of the workgroup; the 3x3/s2 conv (phase 2b)
// is split over K instead of over output rows -- wave (cc, m) keeps the nine W6 fragments of its 32-channel chunk cc and
// m-tile m in REGISTERS for all tiles and accumulates partial sums for all four output rows (four independent MFMA
// chains); the partials meet in LDS (in the L1 tile's storage once every wave is done reading it) and wave (m, r)
// reduces row r.  Four barriers per tile.  The fp32 summation order of phase 2b (four partial sums) differs from the
// layer-wise kernel's, everything else is the same arithmetic.
__global__ __launch_bounds__(512) void bneck_trans_kernel(FusedArgs a) {
    prio_young_half();
    constexpr int IH = 10, IW = 34, NPIX = IH * IW;            // 340 halo pixels
    constexpr int NT1 = 22;
    constexpr int W1_U = 3 * 8 * 64, W5_U = 4 * 9 * 64;         // 16-byte units
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* s_l1 = (bf16_t*)smem;                               // [340][128]  87,040 B  (phase-2b partial sums alias its first 32 KB)
    bf16_t* s_w1 = s_l1 + NPIX * 128;                           // 24,576 B resident
    bf16_t* s_w5 = s_w1 + W1_U * 8;                             // 36,864 B resident
    float* s_b1 = (float*)(s_w5 + W5_U * 8);                    // 512 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // (as a scalar -- readfirstlane -- the wave-dependent loops become branches: measured +3 ... 5 %)
    const int n = lane & 15, g = lane >> 4;
    StageRegs<W1_U> w1regs; StageRegs<W5_U> w5regs;
    stage_load_512<W1_U>(w1regs, a.w1, tid);            // stored to LDS after the first tile's loads have been issued (below)
    stage_load_512<W5_U>(w5regs, a.w5, tid);
    const float b1v = tid < 128 ? a.b1[tid] : 0.f;
    const int cc = wave & 3, m6 = wave >> 2;
    bf16x8 af6[9];
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) af6[s9] = *(const bf16x8*)(a.w6 + (((cc * 9 + s9) * 2 + m6) * 64 + lane) * 8);
    const f32x4 bias6 = *(const f32x4*)(a.b6 + g * 8 + m6 * 4);
    const f32x4 b5 = *(const f32x4*)(a.b5 + g * 4);
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    u32x4 pb[3][3];
    bool p_in[3];
    // the lane's three halo pixels as (row << 8 | column), one register each, unpacked inside issue_pix behind an opaque copy: left to
    // itself the compiler hoists the six quotients / remainders out of the tile loop and, at 256 registers, spills them -- and a
    // spill's reload inside issue_pix is a scratch load whose s_waitcnt vmcnt(0) drains the stores in front of it
    unsigned pyx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int pix = (wave + 8 * t) * 16 + n;
        pix = pix < NPIX ? pix : NPIX - 1;
        pyx[t] = (unsigned)((pix / IW) << 8 | (pix % IW));
    }
    // byte offset of the lane's 16-byte unit of pixel group t from the tile's first halo pixel in the 32-channel source (twice that, plus
    // 64 per chunk, in the 64-channel one): the same for every tile (see bb_chain_kernel)
    unsigned poff[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) poff[t] = (unsigned)(((pyx[t] >> 8) * a.W + (pyx[t] & 255u)) * 64 + g * 16);
    auto issue_pix = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int gy0 = (tt / a.tiles_x) * 8 - 1, gx0 = (tt % a.tiles_x) * 32 - 1;
#ifndef TTUP_NO_FAST_PREFETCH
        if (gy0 >= 0 && gy0 + IH <= a.H && gx0 >= 0 && gx0 + IW <= a.W) {          // halo tile inside the image: scalar bases + lane constants
            const size_t gp0 = (size_t)(b * a.H + gy0) * a.W + gx0;
            const char* base_a = (const char*)(a.a2 + gp0 * 32);
            const char* base_t = (const char*)(a.t2 + gp0 * 64);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                p_in[t] = wave + 8 * t < NT1;          // (groups past the tile: clamped to its last pixel, loaded and never used)
                const unsigned o = opaque_u32(poff[t]), o2 = o * 2u - (unsigned)(g * 16);
                pb[t][0] = *(const u32x4*)(base_a + o);
                pb[t][1] = *(const u32x4*)(base_t + o2);
                pb[t][2] = *(const u32x4*)(base_t + o2 + 64);
            }
            return;
        }
#endif
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int j = wave + 8 * t;
            unsigned q = pyx[t];
            asm volatile("" : "+v"(q));
            const int gy = gy0 + (int)(q >> 8), gx = gx0 + (int)(q & 255u);
            p_in[t] = j < NT1 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const size_t gp = (size_t)(b * a.H + gy) * a.W + gx;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pb[t][c] = u32x4{0u, 0u, 0u, 0u};
                if (p_in[t]) pb[t][c] = (c == 0) ? *(const u32x4*)(a.a2 + gp * 32 + g * 8) : *(const u32x4*)(a.t2 + gp * 64 + (c - 1) * 32 + g * 8);
            }
        }
    };
    if (my_tiles <= 0) return;          // (workgroup-uniform; the launcher never starts more workgroups than tiles)
    issue_pix(0);
    stage_store_512<W1_U>(s_w1, w1regs, tid);
    stage_store_512<W5_U>(s_w5, w5regs, tid);
    if (tid < 128) s_b1[tid] = b1v;
    // every path into the tile loop has the prefetch registers COMPLETE (here: the first tile's; inside the loop: prefetch_arrived in
    // front of phase 2a's stores) -- a path on which they might be pending would put an s_waitcnt vmcnt(0) at the top of every tile
    prefetch_arrived(pb[0]); prefetch_arrived(pb[1]); prefetch_arrived(pb[2]);

    for (int it = 0; it < my_tiles; ++it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * 8, ox0 = (tt % a.tiles_x) * 32;
        TTUP_STAMP_IT(1, it, 0);
        __syncthreads();            // previous tile's reduction has read its partial sums (weights visible on the first pass)
        TTUP_STAMP_IT(1, it, 1);
        // ---------------- phase 1: layer1 halo tile.  Output-channel pairs outermost: a weight fragment read from LDS serves all
        // (up to three) pixel groups of the wave -- 24 fragment reads per wave and tile instead of 72 (the kernel is LDS-bound);
        // every accumulator still sums its three K chunks in the same order
        // (pipelined like conv64_tile_mfma: the two weight fragments of step (q, chunk) + 1 -- and the next pair's bias -- are requested
        // before the MFMAs of step (q, chunk); -DTTUP_NO_FRAG_PIPELINE: each step reads its own)
        bf16x8 afp[2][2];
        f32x4 bqp[2][2];
        auto load_w1 = [&](int st, bf16x8 (&a2)[2]) __attribute__((always_inline)) {          // st = q * 3 + chunk
            const int q = st / 3, chunk = st % 3;
            a2[0] = *(const bf16x8*)(s_w1 + ((chunk * 8 + 2 * q) * 64 + lane) * 8);
            a2[1] = *(const bf16x8*)(s_w1 + ((chunk * 8 + 2 * q + 1) * 64 + lane) * 8);
        };
        auto load_bq = [&](int q, f32x4 (&b2)[2]) __attribute__((always_inline)) {
            b2[0] = *(const f32x4*)(s_b1 + g * 32 + q * 8); b2[1] = *(const f32x4*)(s_b1 + g * 32 + q * 8 + 4);
        };
        load_w1(0, afp[0]);
        load_bq(0, bqp[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 acc[3][2];
            const f32x4 bq0 = bqp[q & 1][0], bq1 = bqp[q & 1][1];
#pragma unroll
            for (int t = 0; t < 3; ++t) { acc[t][0] = bq0; acc[t][1] = bq1; }
#pragma unroll
            for (int chunk = 0; chunk < 3; ++chunk) {
                const int st = q * 3 + chunk;
#ifdef TTUP_NO_FRAG_PIPELINE
                load_w1(st, afp[st & 1]);
                if (chunk == 0) load_bq(q, bqp[q & 1]);
#else
                if (st + 1 < 12) load_w1(st + 1, afp[(st + 1) & 1]);
                if (chunk == 0 && q + 1 < 4) load_bq(q + 1, bqp[(q + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#endif
                const bf16x8 af0 = afp[st & 1][0], af1 = afp[st & 1][1];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (wave + 8 * t >= NT1) continue;             // wave-uniform: waves 6 and 7 own two groups
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, pb[t][chunk]);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr, acc[t][1], 0, 0, 0);
                }
#ifndef TTUP_NO_FRAG_PIPELINE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int j = wave + 8 * t, pix = j * 16 + n;
                if (j >= NT1 || pix >= NPIX) continue;
                const bool inside = p_in[t];
                u32x4 pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = relu_pk(pack2(acc[t][i >> 1][2 * (i & 1)], acc[t][i >> 1][2 * (i & 1) + 1]));
                    pk[i] = inside ? w : 0u;          // (a wave-uniform "interior tile" branch around these selects measured +1 % here, -2 % in the stem)
                }
                *(u32x4*)(s_l1 + l1_off(pix, g * 4 + q)) = pk;
            }
        }
        TTUP_STAMP_IT(1, it, 2);
        __syncthreads();
        TTUP_STAMP_IT(1, it, 3);
        if (it + 1 < my_tiles) issue_pix(it + 1);               // next tile's pixel fragments: in flight during phases 2a and 2b
        // ---------------- phase 2a: 3x3 s1 128 -> 16 on the LDS tile.  A wave owns two VERTICALLY adjacent 16-pixel groups
        // (rows 2q, 2q+1 of column half ch): the four input rows they touch are read once per (chunk, tap column) and
        // shared by both outputs -- 4 fragment reads instead of 6.
        {
            const int q2 = wave >> 1, ch = wave & 1;
            f32x4 acc[2] = {b5, b5};
#if defined(TTUP_NO_FRAG_PIPELINE) || defined(TTUP_ABL_2A_NOMFMA) || defined(TTUP_ABL_2A_NOLOAD)
#pragma unroll 2
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    bf16x8 brow[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
#ifdef TTUP_ABL_2A_NOLOAD
                        brow[rr] = __builtin_bit_cast(bf16x8, pb[rr % 3][dx]);          // (timing ablation: registers instead of LDS reads)
#else
                        brow[rr] = *(const bf16x8*)(s_l1 + l1_off((2 * q2 + rr) * IW + ch * 16 + n + dx, c * 4 + g));
#endif
                    }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
#ifdef TTUP_ABL_2A_NOLOAD
                        const bf16x8 af = af6[dy * 3 + dx];
#else
                        const bf16x8 af = *(const bf16x8*)(s_w5 + ((c * 9 + dy * 3 + dx) * 64 + lane) * 8);
#endif
#ifdef TTUP_ABL_2A_NOMFMA
                        asm volatile("" :: "v"(af), "v"(brow[dy]), "v"(brow[dy + 1]));
#else
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, brow[dy], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, brow[dy + 1], acc[1], 0, 0, 0);
#endif
                    }
                }
#else
            // pipelined like conv64_tile_mfma: the seven fragments of (chunk, tap column) group j+1 are requested before the six MFMAs
            // of group j, and a scheduling barrier keeps the requests there (same k order per accumulator): phase 2a 5.2 k -> 4.7 k cycles,
            // the kernel -3 % (round 5).  It needs 28 more registers than the plain loop: with the 48 swizzled fragment addresses hoisted out
            // of the tile loop the kernel spilled lane constants of issue_pix, whose reloads (scratch loads) put an s_waitcnt vmcnt(0)
            // behind the tile's stores -- hence the opaque column below
            bf16x8 brow[2][4], af[2][3];
            // Swizzled fragment addresses from NINE lane constants instead of 48: pixel P0 + rr * 34 + dx has (pixel & 15) = (P0 + t) & 15
            // with t = 2 rr + dx (34 = 2 mod 16), and chunk (4 c + g) ^ (pixel & 15) = (g ^ (pixel & 15)) ^ (c << 2): the byte address is
            // (bt[t] ^ (c << 6)) + (rr * 34 + dx) * 256 with bt[t] = P0 * 256 + ((g ^ ((P0 + t) & 15)) << 4) -- one v_xor per read, the
            // rest an instruction immediate.  (Written out through l1_off the compiler either hoists 48 addresses out of the tile loop,
            // which spills, or recomputes each with five integer instructions: +240 vector instructions per tile in a kernel whose
            // vector issue port is as busy as its matrix pipe.)
            unsigned bt[9];
            {
                const int P0 = (2 * q2) * IW + ch * 16 + n;
#pragma unroll
                for (int t = 0; t < 9; ++t) bt[t] = (unsigned)(P0 * 256) + (unsigned)(((g ^ ((P0 + t) & 15)) & 15) << 4);
            }
            auto load_group = [&](int j, bf16x8 (&br)[4], bf16x8 (&a3)[3]) __attribute__((always_inline)) {
                const int c = j / 3, dx = j % 3;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) br[rr] = *(const bf16x8*)((const char*)s_l1 + (bt[2 * rr + dx] ^ (unsigned)(c << 6)) + (rr * IW + dx) * 256);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) a3[dy] = *(const bf16x8*)(s_w5 + ((c * 9 + dy * 3 + dx) * 64 + lane) * 8);
            };
            load_group(0, brow[0], af[0]);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                if (j + 1 < 12) load_group(j + 1, brow[(j + 1) & 1], af[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j & 1][dy], brow[j & 1][dy], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j & 1][dy], brow[j & 1][dy + 1], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = oy0 + 2 * q2 + t, ox = ox0 + ch * 16 + n;
                if (oy < a.H && ox < a.W)
                    *(u32x2*)(a.b0 + ((size_t)(b * a.H + oy) * a.W + ox) * 16 + g * 4) =
                        u32x2{relu_pk(pack2(acc[t][0], acc[t][1])), relu_pk(pack2(acc[t][2], acc[t][3]))};
            }
        }
        TTUP_STAMP_IT(1, it, 4);
        // ---------------- phase 2b: 3x3 s2 128 -> 32, K-chunk cc / m-tile m6 of all four output rows
        f32x4 part[4];
        {
            const f32x4 seed = cc == 0 ? bias6 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] = seed;
#pragma unroll
            for (int s9 = 0; s9 < 9; ++s9) {
                const int dy = s9 / 3, dx = s9 % 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pix = (2 * r + dy) * IW + 2 * n + dx;
                    const bf16x8 bfr = *(const bf16x8*)(s_l1 + l1_off(pix, cc * 4 + g));
                    part[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af6[s9], bfr, part[r], 0, 0, 0);
                }
            }
        }
        TTUP_STAMP_IT(1, it, 5);
        __syncthreads();            // every wave is done reading the L1 tile: its storage now carries the partial sums
        TTUP_STAMP_IT(1, it, 6);
        {
            float* s_part = (float*)s_l1;
#pragma unroll
            for (int r = 0; r < 4; ++r) *(f32x4*)(s_part + (((m6 * 4 + cc) * 4 + r) * 64 + lane) * 4) = part[r];
        }
        __syncthreads();
        {
            const float* s_part = (const float*)s_l1;
            const int mr = wave >> 2, rr = wave & 3;            // this wave reduces m-tile mr, output row rr
            f32x4 v = *(const f32x4*)(s_part + (((mr * 4 + 0) * 4 + rr) * 64 + lane) * 4);
#pragma unroll
            for (int c = 1; c < 4; ++c) v += *(const f32x4*)(s_part + (((mr * 4 + c) * 4 + rr) * 64 + lane) * 4);
            const int OH = (a.H + 1) >> 1, OW = (a.W + 1) >> 1;
            const int oy = (oy0 >> 1) + rr, ox = (ox0 >> 1) + n;
            // The next tile's pixel fragments (requested at the start of phase 2a) are waited for HERE, in front of the tile's LAST stores:
            // at the top of the next tile, behind them, the wait is an s_waitcnt vmcnt(0) that drains those stores as well
            // (prefetch_arrived; unconditional: behind a branch the compiler would wait again at the top).  Not earlier: under load a
            // read takes ~5 k cycles to come back (phase stamps, round 5: phase 2a lasted 5.2 k cycles with or without its MFMAs and LDS
            // reads while the wait stood at its end) -- phases 2a, 2b and the two barriers together cover that, phase 2a alone does not.
            prefetch_arrived(pb[0]); prefetch_arrived(pb[1]); prefetch_arrived(pb[2]);
            if (oy < OH && ox < OW)
                *(u32x2*)(a.b1o + ((size_t)(b * OH + oy) * OW + ox) * 32 + g * 8 + mr * 4) = u32x2{relu_pk(pack2(v[0], v[1])), relu_pk(pack2(v[2], v[3]))};
        }
    }
}

int launch_bneck_trans(const PackedConv& p1, const PackedConv& p5, const PackedConv& p6, const void* a2, const void* t2,
                       void* b0, void* b1, int batch, int h, int w, hipStream_t st) {
    TTUP_REQUIRE(p1.cout == 128 && p1.cin_total == 96 && p1.c0 == 32 && p1.k == 1 && p1.ck == 32, TTUP_EINVAL, "bneck_trans: unexpected conv1 shape");
    TTUP_REQUIRE(p5.cout == 16 && p5.cin_total == 128 && p5.k == 3 && p5.stride == 1 && p5.ck == 32, TTUP_EINVAL, "bneck_trans: unexpected conv5 shape");
    TTUP_REQUIRE(p6.cout == 32 && p6.cin_total == 128 && p6.k == 3 && p6.stride == 2 && p6.ck == 32, TTUP_EINVAL, "bneck_trans: unexpected conv6 shape");
    TTUP_REQUIRE(h % 2 == 0 && w % 2 == 0, TTUP_EINVAL, "bneck_trans: even input size required");
    FusedArgs a;
    a.a2 = (const bf16_t*)a2; a.t2 = (const bf16_t*)t2;
    a.w1 = (const bf16_t*)p1.w_dev; a.b1 = p1.bias_dev; a.w5 = (const bf16_t*)p5.w_dev; a.b5 = p5.bias_dev;
    a.w6 = (const bf16_t*)p6.w_dev; a.b6 = p6.bias_dev; a.b0 = (bf16_t*)b0; a.b1o = (bf16_t*)b1;
    a.H = h; a.W = w; a.tiles_x = cdiv(w, 32); a.tiles_per_img = a.tiles_x * cdiv(h, 8); a.total_tiles = a.tiles_per_img * batch;
    constexpr size_t SMEM = (size_t)(340 * 128 + 3 * 8 * 64 * 8 + 4 * 9 * 64 * 8) * 2 + 512;
    if (int rc = ensure_max_lds((const void*)bneck_trans_kernel, SMEM)) return rc;
    const int grid = a.total_tiles < 256 ? a.total_tiles : 256;
    if (grid == 0) return TTUP_OK;
    kernel_note("bneck_trans_kernel");
    hipLaunchKernelGGL(bneck_trans_kernel, dim3(grid), dim3(512), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// ------------------------------------------------------------------ fused BasicBlock chains
// NB BasicBlocks (wasb.py:48-64: conv3x3+BN+ReLU, conv3x3+BN, +x, ReLU) of one HRNet branch in ONE kernel.
// The input tile with a 2*NB-pixel halo is staged once; every intermediate (rounded to bf16 exactly like the unfused
// path, and zeroed outside the image so that each conv sees its own zero padding) lives in LDS; only the final
// TH x TW tile is written.  HBM traffic per block chain: one read + one write of the tensor instead of 5 passes per block.
// Each wave keeps the conv's A fragments (weights) in registers and walks 16-pixel groups of the output region
// (linear pixel index, so ragged region widths waste nothing).
struct BBArgs {
    const bf16_t* x; bf16_t* y;
    const bf16_t* w[4]; const float* bias[4];
    int H, W, tiles_x, tiles_per_img, total_tiles;
    // optional 1x1 follower on the chain output (C=32 -> 16, BN folded, no ReLU: the fuse-layer conv of wasb.py:189-205 that
    // feeds the higher-resolution branch): one extra MFMA per 16-pixel group on the bf16 pairs just packed
    const bf16_t* wf; const float* bf; bf16_t* yf;
    // C=16 two-block chain at full resolution: the fuse-layer sum that consumes the branch (wasb.py:236-243) rides in the last
    // conv's epilogue: ysum = relu(y + sum_k up(st[k], 2^ssh[k])).  With `heat` set the sum is the stage-4 output: it is not
    // stored at all, the 1x1 head (final_layers[0] channel 1, wasb.py:484,606) is applied to it in registers and the workgroup
    // leaves its argmax partial (pv/pi[map * nblk + tile]); y itself (the pre-fuse branch tensor) is only stored when a.y is set.
    const bf16_t* st[3]; int ssh[3]; int nsum; bf16_t* ysum;
    float* heat; const float* hw; float hbias; float* pv; long long* pi;
};
// the tile's slices of the fuse-layer terms staged in LDS by the chain kernel (element offset of term k, pixels per row); a
// separate by-value struct: writing into the kernel-argument struct would move all of it to scratch memory
struct BBTermLds { const bf16_t* s_terms; int toff[3]; int tw[3]; };

// ReLU on the sign bit (one integer max, like relu_pk on bf16 pairs): negative values and -0 become +0, +NaN stays NaN
__device__ __forceinline__ float relu_f32(float v) { const int b = __float_as_int(v); return __int_as_float(b > 0 ? b : 0); }
struct BBBest { float v; long long i; };
__device__ __forceinline__ bool bb_better(float v, long long i, float bv, long long bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

// (value, index) as one unsigned key: greater key = greater value (NaN greatest, -0 == +0), then lower index (index < 2^31)
__device__ __forceinline__ unsigned long long bb_key(float v, int e) {
    v += 0.0f;                                              // -0 -> +0
    const unsigned bits = __float_as_uint(v);
    unsigned k = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
    if (v != v) k = 0xffffffffu;
    return ((unsigned long long)k << 32) | (unsigned)(~e);
}
__device__ __forceinline__ float bb_key_value(unsigned long long key) {
    const unsigned k = (unsigned)(key >> 32);
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// lane i <- lane i + n of the same 16-lane row (0 where that lane does not exist): DPP row_shl, no LDS traffic
__device__ __forceinline__ unsigned long long bb_dpp_shl(unsigned long long x, int n) {
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    switch (n) {
        case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x108, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x108, 0xf, 0xf, false); break;
        case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0xf, false); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x102, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x102, 0xf, 0xf, false); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, false); break;
    }
    return ((unsigned long long)hi << 32) | lo;
}

// element offset of 8-channel chunk c8 of the pixel at buffer column x (pix = row*stride + x); C=32 swizzles the chunk
// with bits 1..2 of the column (conflict-free ds_read_b128, see lds_off).  C=16 (32 B per pixel) flips the two chunks with
// bit 2 of the column: the ds_read_b128 fragments of a band group stay conflict-free (columns x and x+8 of a hardware lane group
// carry different chunks either way) and the epilogue's 8-byte stores, 16 lanes at a 32-byte stride, are 2-way instead of 4-way.
// (History of SQ_LDS_BANK_CONFLICT per launch of the two-block chain: 22 % of its LDS cycles before this swizzle, 6.45e6 = 10 %
// with it in round 2, 1.38e7 = 22 % again in round 3 when the ragged strips were packed row-major across aliasing rows, and back
// down with odd row strides + column strip groups in round 4: profiles/r4_pmc_summary.txt.)
template <int C> __device__ __forceinline__ int bb_off(int pix, int x, int c8) {
    if (C == 32) return pix * 32 + ((c8 ^ ((x >> 1) & 3)) << 3);
    return pix * C + ((c8 ^ ((x >> 2) & 1)) << 3);
}

// Weight fragments + bias of one 16-channel conv, loaded by the CALLER: the chain kernel requests the next conv's fragments from
// L2 before the barrier that ends the current conv, so their latency (the first MFMA of a conv needs all of them) hides behind
// the barrier wait instead of following it.
struct BBFrag16 { bf16x8 af[5]; f32x4 bias; };
// A fragment of the 16x16 identity for lanes g >= 2 (row n, columns (g & 1) * 8 .. + 7): the residual add of a block's second
// conv rides in the unused half of its last k-step.  Built once per kernel (it costs ~35 vector instructions).
__device__ __forceinline__ bf16x8 bb_identity_frag(int lane) {
    const int n = lane & 15, g = lane >> 4;
    unsigned short idm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) idm[j] = (n == (g & 1) * 8 + j) ? 0x3F80 : 0;
    return __builtin_bit_cast(bf16x8, idm);
}
// K order of the 16-channel chain convs (two taps per k-step, first tap on lane groups 0-1, second on 2-3):
//   (0,0)|(0,1)   (1,0)|(1,1)   (2,0)|(2,1)   (0,2)|(1,2)   (2,2)|pad
// so that the pixel fragment of the first three steps depends on the input row only (row y+dy, columns x | x+1): a wave walking
// consecutive output rows reads it once for three rows, and the fourth step's fragment (column x+2 of rows r | r+1) doubles as
// the fifth step of the row two above.  The weights stay in the standard packing (taps 2s | 2s+1 per step): tap t of lane group
// half c8 is at step t/2, lane group 2(t&1) + c8 -- a gather at load time, no second packing.
__device__ __forceinline__ int bb_tap16(int s, int h) {          // tap of k-step s, half h (9 = the zero pad)
    return s < 3 ? 3 * s + h : (s == 3 ? (h ? 5 : 2) : (h ? 9 : 8));
}
__device__ __forceinline__ bf16x8 bb_weight_frag16(const bf16_t* wfrag, int s, int lane) {
    const int i = lane & 15, g = lane >> 4, tap = bb_tap16(s, g >> 1);
    return *(const bf16x8*)(wfrag + ((tap >> 1) * 64 + i + 16 * (2 * (tap & 1) + (g & 1))) * 8);
}
__device__ __forceinline__ void bb_load_frag16(BBFrag16& f, const bf16_t* wfrag, const float* biasp, int lane) {
#pragma unroll
    for (int s = 0; s < 5; ++s) f.af[s] = bb_weight_frag16(wfrag, s, lane);
    f.bias = *(const f32x4*)(biasp + (lane >> 4) * 4);
}

// One 3x3 conv of the chain.  Input buffer: row stride RWI pixels, region origin at (IOFF,IOFF).  Output region RHO x RWO.
// SECOND: second conv of a BasicBlock -> adds the block input (buffer s_res, row stride RWR, origin offset ROFF) and the
// result either overwrites that buffer in place (ORW = RWR, OOFF = ROFF: each pixel is read and written by the same lane)
// or goes to global memory.  A wave owns whole output rows (y = wave, wave+8, ...); the 16-pixel groups of a row are
// unrolled so every LDS address is a per-lane base plus an immediate.
// The last conv of the C=16 chain reads its fuse-sum / head configuration from BBArgs at run time here; the forms the network uses are
// compiled out in csrc/chain16.h (c16_chain_kernel), this one is the fallback for other term layouts and the cross-check of those.
template <int R> struct BBRow { static constexpr int value = R; };
// NWV (C=32 only): waves that share the conv's rows -- `wave` is the wave's index among them (rows wave, wave + NWV, ...).  af32: the C=32
// conv's 18 weight fragments already in registers (a two-group variant kept them there for the life of the workgroup: csrc/experiments).
template <int C, int RWI, int IOFF, int RHO, int RWO, bool SECOND, int RWR, int ROFF, bool GLOBAL_OUT, int ORW, int OOFF, int NWV = 8>
__device__ __forceinline__ void bb_conv(const bf16_t* s_in, bf16_t* s_out, const bf16_t* s_res, const bf16_t* wfrag, const float* biasp,
                                        bf16_t* gout, int gy0, int gx0, int H, int W, int b, int wave, int lane,
                                        const bf16_t* wf = nullptr, const float* bfp = nullptr, bf16_t* yf = nullptr,
                                        const BBArgs* ex = nullptr, BBBest* best = nullptr, const BBFrag16* pre = nullptr,
                                        const BBTermLds* tl = nullptr, bf16x8 idm_pre = bf16x8{}, const bf16x8* af32 = nullptr) {
    constexpr int MT = C / 16;
    constexpr int KSTEPS = (C == 16) ? 5 : 9;
    constexpr int XT = (RWO + 15) / 16;
    static_assert(NWV == 8 || C == 32, "only the 32-channel row loop takes a wave count");
    const int n = lane & 15, g = lane >> 4;
    bf16x8 af[KSTEPS][MT];
    if (C == 32 && af32) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) af[s][m] = af32[s * MT + m];
    } else if (C == 16 && pre) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) af[s][0] = pre->af[s < 5 ? s : 4];
    } else {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) af[s][m] = (C == 16) ? bb_weight_frag16(wfrag, s < 5 ? s : 4, lane) : *(const bf16x8*)(wfrag + ((s * MT + m) * 64 + lane) * 8);
    }
    // C=16, second conv of a block: the unused tenth tap of the last k-step (lanes g >= 2, zero weights) carries the block
    // input through an identity matrix, so the residual add happens inside the MFMA (exact: bf16 * 1.0 into the fp32 sum)
    constexpr bool RES_MFMA = SECOND && C == 16;
    if (RES_MFMA && g >= 2) af[KSTEPS - 1][0] = pre ? idm_pre : bb_identity_frag(lane);      // (by value: a field of *pre would pin the struct in memory)
    f32x4 bias[MT];
    if (C == 16 && pre) bias[0] = pre->bias;
    else {
#pragma unroll
        for (int m = 0; m < MT; ++m) bias[m] = *(const f32x4*)(biasp + g * 4 * MT + m * 4);
    }
    int koff[KSTEPS];                     // C=32: per-lane tap/channel offset of every k-step (elements); k-step s = tap (s/3, s%3)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int dy = s / 3, dx = s % 3;
        // the column swizzle depends only on (n + dx + IOFF) mod 8: 16-pixel groups start at multiples of 16
        koff[s] = (dy * RWI + dx) * C + ((g ^ (((n + dx + IOFF) >> 1) & 3)) << 3);
    }
    // lane's pixel in the last (possibly ragged) group is clamped so that reads stay inside the buffer
    constexpr int XLAST = (XT - 1) * 16;
    const int nl = (XLAST + n < RWO) ? n : (RWO - 1 - XLAST);
    // lane's first output channel inside its pixel record (chunk g for C=32, chunk g>>1 + half g&1 for C=16; swizzled like bb_off)
    const int res_ch = (C == 32) ? ((g ^ (((n + ROFF) >> 1) & 3)) << 3) : ((((g >> 1) ^ (((n + ROFF) >> 2) & 1)) << 3) + (g & 1) * 4);
    const int out_ch = (C == 32) ? ((g ^ (((n + OOFF) >> 1) & 3)) << 3) : ((((g >> 1) ^ (((n + OOFF) >> 2) & 1)) << 3) + (g & 1) * 4);
    // zero padding of the next conv: outputs outside the image must be 0; only border tiles have any (wave-uniform test)
#ifdef TTUP_ABL_NOPAD
    const bool interior = true;
#else
    const bool interior = gy0 >= 0 && gy0 + RHO <= H && gx0 >= 0 && gx0 + RWO <= W;
#endif
    constexpr bool CAN_FOLLOW = GLOBAL_OUT && C == 32;
    bf16x8 af_f = {};
    f32x4 bias_f = {0.f, 0.f, 0.f, 0.f};
    if (CAN_FOLLOW && yf) { af_f = *(const bf16x8*)(wf + lane * 8); bias_f = *(const f32x4*)(bfp + g * 4); }
    constexpr bool CAN_SUM = GLOBAL_OUT && C == 16;
    f32x4 hw4 = {0.f, 0.f, 0.f, 0.f};
    if (CAN_SUM && ex && ex->heat) hw4 = *(const f32x4*)(ex->hw + g * 4);
    // C=16: a wave owns a BAND of consecutive output rows (pixel fragments shared between them, see bb_tap16); C=32: rows
    // wave, wave+8, ... (two output tiles per fragment read already)
    constexpr bool BAND = (C == 16);
    constexpr int RB = (RHO + 7) / 8;
    const int yb = BAND ? wave * RB : wave;      // the wave's first row
    // C=32: per-lane fragment addresses of the wave's FIRST row, one per k-step (full groups / clamped last group); the row loop is
    // fully unrolled, so the rows that follow are compile-time offsets (LDS instruction immediates) from them instead of a
    // dozen address registers that each need an add per row.  (C=16 sets up its band addresses below.)
    const bf16_t* pk0[KSTEPS];
    const bf16_t* pkl[KSTEPS];
    {
        const bf16_t* row0 = s_in + ((wave + IOFF) * RWI + IOFF) * C;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) { pk0[s] = row0 + n * C + koff[s]; pkl[s] = row0 + (XLAST + nl) * C + koff[s]; }
    }
    constexpr int ROWSTEP = NWV * RWI * C;
    bf16_t* const so0 = GLOBAL_OUT ? nullptr : s_out + ((yb + OOFF) * ORW + n + OOFF) * C + out_ch;      // lane's output slot in the wave's first row
    // global stores: wave-uniform row base (scalar registers) + the lane's byte offset inside a 16-pixel group (one register for C-channel
    // records, one for 16-channel records) + the group as an immediate -- instead of a 64-bit per-lane address chain per store
    const unsigned st_c = (unsigned)((n * C + g * 4 * MT) * 2), st_16 = (unsigned)((n * 16 + g * 4) * 2);
    // epilogue of one 16-pixel group of row y (orow = its row offset from the wave's first row): bias/ReLU/rounding, zero padding
    // of the next conv, stores, and whatever rides in the last conv's epilogue
    auto epi = [&](int xt, int orow, int y, const f32x4 (&accx)[MT]) __attribute__((always_inline)) {
        const int gy = gy0 + y;
        const bool row_in = gy >= 0 && gy < H;
        const int x = xt * 16 + n;
        const bool valid = !(xt == XT - 1 && x >= RWO);       // ragged last group: computed (the follower MFMA needs the whole wave), not stored
        float v[4 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[m * 4 + r] = accx[m][r];
        if (SECOND && !RES_MFMA) {       // + block input; the lane's 4*MT channels start at g*4*MT
            const bf16_t* rp = s_res + ((yb + ROFF) * RWR + n + ROFF) * C + res_ch + (orow * RWR + xt * 16) * C;
            const u32x4 rv = *(const u32x4*)rp;
            const unsigned w4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32((bf16_t)(w4[k] & 0xffff)); v[2 * k + 1] += bf16_to_f32((bf16_t)(w4[k] >> 16)); }
        }
        unsigned pk[2 * MT];
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) pk[i] = relu_pk(pack2(v[2 * i], v[2 * i + 1]));
        const int gx = gx0 + x;
        bool inside = true;
        if (!interior) {
            inside = row_in && gx >= 0 && gx < W;
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) pk[i] = inside ? pk[i] : 0u;
        }
        const size_t rowpix = (size_t)(b * H + gy) * W + gx0;          // (wave-uniform) first pixel of the region's row in the image
        if (GLOBAL_OUT) {
            if (inside && valid && gout) {
                char* o = (char*)(gout + rowpix * C) + (opaque_u32(st_c) + (unsigned)(xt * 16 * C * 2));
                if (C == 16) *(u32x2*)o = u32x2{pk[0], pk[1]};
                else *(u32x4*)o = u32x4{pk[0], pk[1], pk[2], pk[3]};
            }
            if constexpr (CAN_SUM) {
                if (ex && (ex->nsum > 0 || ex->heat)) {
                    // fuse-layer sum on the rounded block output, exactly what the element-wise pass read back from memory
                    float ys[4] = {bf16_to_f32((bf16_t)(pk[0] & 0xffff)), bf16_to_f32((bf16_t)(pk[0] >> 16)),
                                   bf16_to_f32((bf16_t)(pk[1] & 0xffff)), bf16_to_f32((bf16_t)(pk[1] >> 16))};
                    // stage-4 tail: neither the branch tensor nor the sum is stored, so neither is rounded to bf16 -- the head
                    // sees the fp32 values (two roundings fewer right in front of the heatmap: a smaller bf16-path error)
                    const bool exact_tail = ex->heat && !gout && !ex->ysum;
                    if (exact_tail) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ys[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                    const bool live = inside && valid;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (k >= ex->nsum) continue;
                        // the tile's slice of term k sits in LDS (staged during the previous conv): no memory round trip here
                        const int sh = ex->ssh[k];
                        const u32x2 tv = *(const u32x2*)(tl->s_terms + tl->toff[k] + ((((gy0 + y) >> sh) - (gy0 >> sh)) * tl->tw[k] + ((gx >> sh) - (gx0 >> sh))) * 16 + g * 4);
                        ys[0] += bf16_to_f32((bf16_t)(tv.x & 0xffff)); ys[1] += bf16_to_f32((bf16_t)(tv.x >> 16));
                        ys[2] += bf16_to_f32((bf16_t)(tv.y & 0xffff)); ys[3] += bf16_to_f32((bf16_t)(tv.y >> 16));
                    }
                    const unsigned q0 = relu_pk(pack2(ys[0], ys[1])), q1 = relu_pk(pack2(ys[2], ys[3]));
                    if (ex->ysum && live) *(u32x2*)(ex->ysum + ((size_t)(b * H + gy) * W + gx) * 16 + g * 4) = u32x2{q0, q1};
                    if (ex->heat) {
                        // head on the bf16-rounded sum: this lane's 4 channels, then across the 4 lane groups of the pixel
                        float hy[4] = {bf16_to_f32((bf16_t)(q0 & 0xffff)), bf16_to_f32((bf16_t)(q0 >> 16)),
                                       bf16_to_f32((bf16_t)(q1 & 0xffff)), bf16_to_f32((bf16_t)(q1 >> 16))};
                        if (exact_tail) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) hy[r] = ys[r] > 0.f ? ys[r] : 0.f;
                        }
                        float part = hy[0] * hw4[0];
                        part = fmaf(hy[1], hw4[1], part);
                        part = fmaf(hy[2], hw4[2], part);
                        part = fmaf(hy[3], hw4[3], part);
                        part += __shfl_xor(part, 16, 64);
                        part += __shfl_xor(part, 32, 64);
                        const float hv = part + ex->hbias;
                        if (live && g == 0) {
                            const long long e = (long long)gy * W + gx;
                            ex->heat[(size_t)b * H * W + e] = hv;
                            if (bb_better(hv, e, best->v, best->i)) { best->v = hv; best->i = e; }
                        }
                    }
                }
            }
            if constexpr (CAN_FOLLOW) {
                if (yf) {          // lane (n, g) holds channels 8g..8g+7 of its pixel = k-group g of the follower's only k-step
                    const u32x4 bq = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    const f32x4 cf = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_f, __builtin_bit_cast(bf16x8, bq), bias_f, 0, 0, 0);
                    if (inside && valid) *(u32x2*)((char*)(yf + rowpix * 16) + (opaque_u32(st_16) + (unsigned)(xt * 16 * 32))) = u32x2{pack2(cf[0], cf[1]), pack2(cf[2], cf[3])};
                }
            }
        } else if (valid) {
            bf16_t* o = so0 + (orow * ORW + xt * 16) * C;
            if (C == 16) *(u32x2*)o = u32x2{pk[0], pk[1]};
            else *(u32x4*)o = u32x4{pk[0], pk[1], pk[2], pk[3]};
        }
    };
    if constexpr (!BAND) {
        // (the run-time epilogue form, MODE 0 with the fuse sum, is a cross-check path and stays rolled: unrolled it spills)
        constexpr int ROW_UNROLL = (RHO + NWV - 1) / NWV;
#pragma unroll ROW_UNROLL
        for (int yj = 0; yj < (RHO + NWV - 1) / NWV; ++yj) {
            const int y = wave + NWV * yj;
            if (y >= RHO) break;
            f32x4 acc[XT][MT];
#pragma unroll
            for (int xt = 0; xt < XT; ++xt)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[xt][m] = bias[m];
#ifdef TTUP_NO_FRAG_PIPELINE
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                bf16x8 bfr[XT];
#pragma unroll
                for (int xt = 0; xt < XT; ++xt)
                    bfr[xt] = (xt < XT - 1) ? *(const bf16x8*)(pk0[s] + yj * ROWSTEP + xt * 16 * C) : *(const bf16x8*)(pkl[s] + yj * ROWSTEP);
#pragma unroll
                for (int xt = 0; xt < XT; ++xt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[xt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][m], bfr[xt], acc[xt][m], 0, 0, 0);
            }
#else
            // pipelined (see conv64_tile_mfma): the pixel fragments of k-step s+1 are requested before the MFMAs of step s
            bf16x8 bfr[2][XT];
            auto load_step = [&](int s, bf16x8 (&bf)[XT]) __attribute__((always_inline)) {
#pragma unroll
                for (int xt = 0; xt < XT; ++xt)
                    bf[xt] = (xt < XT - 1) ? *(const bf16x8*)(pk0[s] + yj * ROWSTEP + xt * 16 * C) : *(const bf16x8*)(pkl[s] + yj * ROWSTEP);
            };
            load_step(0, bfr[0]);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (s + 1 < KSTEPS) load_step(s + 1, bfr[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int xt = 0; xt < XT; ++xt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[xt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][m], bfr[s & 1][xt], acc[xt][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) epi(xt, NWV * yj, y, acc[xt]);
        }
    } else {
        const int h = g >> 1, c8 = g & 1;
        constexpr int RS = RWI * C;                                   // one input row (elements)
        constexpr bool RAGGED = RWO % 16 != 0;
        // element offset of the lane's 8-channel chunk of the pixel dx columns right of output pixel nn (swizzle as in bb_off)
        auto lane_off = [&](int nn, int dx) { return (nn + dx) * C + ((c8 ^ (((nn + dx + IOFF) >> 2) & 1)) << 3); };
        const bf16_t* rowb = s_in + ((yb + IOFF) * RWI + IOFF) * C;
        const bf16_t* pA = rowb + lane_off(n, h);                      // steps 0-2: row r + dy, column x | x+1
        const bf16_t* pAl = rowb + XLAST * C + lane_off(nl, h);
        const bf16_t* pC = rowb + h * RS + lane_off(n, 2);             // step 3: column x+2 of rows r | r+1
        const bf16_t* pCl = rowb + XLAST * C + h * RS + lane_off(nl, 2);
        // step 4: pixel (r+2, x+2) on the first half; second half: the block input at the output pixel (second conv of a block,
        // identity weights) or the same pixel again (zero weights)
        const bf16_t* pD = rowb + 2 * RS + lane_off(n, 2);
        const bf16_t* pDl = rowb + XLAST * C + 2 * RS + lane_off(nl, 2);
        int dstep = RS;
        if (RES_MFMA && h) {
            const bf16_t* rr = s_res + ((yb + ROFF) * RWR + ROFF) * C;
            pD = rr + n * C + ((c8 ^ (((n + ROFF) >> 2) & 1)) << 3);
            pDl = rr + (XLAST + nl) * C + ((c8 ^ (((nl + ROFF) >> 2) & 1)) << 3);
            dstep = RWR * C;
        }
        // Rows of the band one after the other, the row's XT column groups as independent accumulator chains (as in the 32-channel
        // form).  Per row and group: ONE new fragment for steps 0-2 (row r+2; rows r and r+1 are still in registers from the rows
        // before) plus the fragments of steps 3 and 4 -- three LDS reads for five MFMAs instead of five.
        // Ragged region widths (38 / 36 / 34 px = two full 16-pixel groups + 6 / 4 / 2 px): the band walks the FULL groups only; the
        // leftover strip (RHO rows x RX columns) is packed 16 pixels at a time into "strip groups" whose lanes sit in different rows
        // -- 12 / 7 / 4 groups instead of 30 / 28 / 26 two-thirds-empty ones -- and handed to the waves with spare time: the last
        // wave's band is short or empty (RHO is not a multiple of 8), so it takes the first K0 strip groups, the others one or two each.
        // Same k-step order and operands per output pixel as a band group: bit-identical results.
#ifdef TTUP_NO_STRIP
        constexpr bool STRIP = false;
#else
        constexpr bool STRIP = RAGGED && !GLOBAL_OUT;
#endif
        constexpr int XTR = STRIP ? XT - 1 : XT;
        if (yb < RHO) {
            bf16x8 fa[XT][RB + 2];
#pragma unroll
            for (int xt = 0; xt < XTR; ++xt) {
                const bf16_t* bA = (RAGGED && xt == XT - 1) ? pAl : pA + xt * 16 * C;
                fa[xt][0] = *(const bf16x8*)bA; fa[xt][1] = *(const bf16x8*)(bA + RS);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int y = yb + r;
                if (y >= RHO) break;
                bf16x8 f3[XT], f4[XT];
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) {
                    const bool lastg = RAGGED && xt == XT - 1;
                    fa[xt][r + 2] = *(const bf16x8*)((lastg ? pAl : pA + xt * 16 * C) + (r + 2) * RS);
                    f3[xt] = *(const bf16x8*)((lastg ? pCl : pC + xt * 16 * C) + r * RS);
                    f4[xt] = *(const bf16x8*)((lastg ? pDl : pD + xt * 16 * C) + r * dstep);
                }
                f32x4 acc[XT][1];
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][0], fa[xt][r], bias[0], 0, 0, 0);
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][0], fa[xt][r + 1], acc[xt][0], 0, 0, 0);
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][0], fa[xt][r + 2], acc[xt][0], 0, 0, 0);
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3][0], f3[xt], acc[xt][0], 0, 0, 0);
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4][0], f4[xt], acc[xt][0], 0, 0, 0);
#pragma unroll
                for (int xt = 0; xt < XTR; ++xt) epi(xt, r, y, acc[xt]);
            }
        }
        if constexpr (STRIP) {
            // Strip groups are COLUMN groups when the input buffer's row stride is odd (the two-block chain's buffers): 16 consecutive
            // rows of one strip column.  A pixel record is two 16-byte LDS slots and a ds_read_b128 is served in lane groups
            // {0-3,12-15 of one chunk | 4-11 of the other}: with a row stride of 1 (mod 8) pixels the 16 rows land on the 16 slots
            // exactly like the 16 consecutive pixels of a band group -- conflict-free for every tap, and the swizzle terms (a
            // function of the column) become wave-uniform.  Round 3's row-major packing (16 pixels over 3-8 rows of a 6 / 4 / 2-pixel
            // strip, row stride 40 = 0 mod 8: rows aliased on the same banks) doubled the kernel's SQ_LDS_BANK_CONFLICT
            // (6.45e6 -> 1.38e7 per launch); it is kept for even strides (the one-block chain).
            constexpr bool COLG = (RWI & 1) == 1 && (!SECOND || (RWR & 1) == 1);
            constexpr int RX = RWO - XLAST, NSP = RHO * RX;
            constexpr int CG = (RHO + 15) / 16;                          // column groups per strip column
            constexpr int NSG = COLG ? RX * CG : (NSP + 15) / 16;
            constexpr int ROWS7 = RHO - 7 * RB < 0 ? 0 : (RHO - 7 * RB > RB ? RB : RHO - 7 * RB);      // band rows of the last wave
            // a strip group costs about two band groups (five fragment reads instead of three, one dependent MFMA chain, per-lane
            // addresses): the last wave takes as many as fit in HALF of its band's gap (in band-group units), the rest go round
#ifdef TTUP_STRIP_K0_FULL
            constexpr int K0 = NSG < 2 * (RB - ROWS7) ? NSG : 2 * (RB - ROWS7);
#else
            constexpr int K0 = NSG < RB - ROWS7 ? NSG : RB - ROWS7;
#endif
            static_assert(NSG - K0 <= 16, "at most two strip groups per wave after the last wave's share");
            auto strip = [&](int j) __attribute__((always_inline)) {
                int row, col;
                bool valid;
                if constexpr (COLG) {
                    const int cj = j / CG, rg = j - cj * CG;              // wave-uniform
                    // rows dealt evenly over the column's groups (30 rows: 15 + 15, not 16 + 14)
                    constexpr int RPG = (RHO + CG - 1) / CG;
                    const int r0 = rg * RPG;
                    valid = n < RPG && r0 + n < RHO;
                    const int rn = r0 + (n < RPG ? n : RPG - 1);         // idle lanes re-read a neighbour's addresses (identical addresses
                    row = rn < RHO ? rn : RHO - 1;                        // broadcast: no bank conflict) and store nothing
                    col = XLAST + cj;
                } else {
                    const int p = 16 * j + n;
                    valid = p < NSP;
                    const int pc = valid ? p : NSP - 1;                   // lanes past the strip recompute its last pixel and store nothing
                    row = pc / RX; col = XLAST + (pc - row * RX);
                }
                const bf16_t* b0 = s_in + ((row + IOFF) * RWI + IOFF + col) * C;
                const int sw2 = (c8 ^ (((col + 2 + IOFF) >> 2) & 1)) << 3;
                const bf16_t* a0 = b0 + h * C + ((c8 ^ (((col + h + IOFF) >> 2) & 1)) << 3);        // steps 0-2: rows row + dy, column col | col+1
                const bf16_t* a3 = b0 + h * RS + 2 * C + sw2;                                       // step 3: column col+2 of rows row | row+1
                const bf16_t* a4 = b0 + 2 * RS + 2 * C + sw2;                                       // step 4: (row+2, col+2) | block input / pad
                if (RES_MFMA && h) a4 = s_res + ((row + ROFF) * RWR + ROFF + col) * C + ((c8 ^ (((col + ROFF) >> 2) & 1)) << 3);
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][0], *(const bf16x8*)a0, bias[0], 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][0], *(const bf16x8*)(a0 + RS), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][0], *(const bf16x8*)(a0 + 2 * RS), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3][0], *(const bf16x8*)a3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4][0], *(const bf16x8*)a4, acc, 0, 0, 0);
                unsigned q0 = relu_pk(pack2(acc[0], acc[1])), q1 = relu_pk(pack2(acc[2], acc[3]));
                if (!interior) {                                      // zero padding of the next conv outside the image
                    const int gy = gy0 + row, gx = gx0 + col;
                    const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    q0 = inside ? q0 : 0u; q1 = inside ? q1 : 0u;
                }
                if (valid) *(u32x2*)(s_out + ((row + OOFF) * ORW + col + OOFF) * C + ((((g >> 1) ^ (((col + OOFF) >> 2) & 1)) << 3) + (g & 1) * 4)) = u32x2{q0, q1};
            };
#ifndef TTUP_ABL_NOSTRIPWORK
            if (wave == 7) {
#pragma unroll
                for (int j = 0; j < K0; ++j) strip(j);
            }
            if (K0 + wave < NSG) strip(K0 + wave);
            if (NSG - K0 > 8 && K0 + 8 + wave < NSG) strip(K0 + 8 + wave);
#endif
        }
    }
}

// Persistent: a workgroup walks tiles; the next tile's input region is prefetched into registers while the current one is
// computed (C=32).  C=32 keeps the weights of both convs (2 x 18 KB) resident in LDS; C=16 runs one tile
// per workgroup with its 5 weight fragments per conv straight from L2 (persistent variants measured slower there).
template <int C, int NB, int TH, int TW>
__global__ __launch_bounds__(512) void bb_chain_kernel(BBArgs a) {
    prio_young_half();
    constexpr int L = 2 * NB;
    constexpr int R0H = TH + 2 * L, R0W = TW + 2 * L;
    constexpr int SZ_A = R0H * R0W * C, SZ_B = (R0H - 2) * (R0W - 2) * C;
    constexpr int KSTEPS = (C == 16) ? 5 : 9, MT = C / 16;
    constexpr int W_UNITS = KSTEPS * MT * 64;                    // 16-byte units per conv
    constexpr bool RESIDENT = (C == 32);                         // both convs' weights (2 x 18 KB) stay in LDS: no rotation, two barriers fewer per tile
    constexpr bool WGLOBAL = (C == 16);                          // C=16: one tile per workgroup, weight fragments straight from global/L2
    constexpr int W_PT = (W_UNITS + 511) / 512;
    constexpr int IN_UNITS = R0H * R0W * (C / 8), IN_PT = (IN_UNITS + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* bufA = (bf16_t*)smem;              // block input region (later overwritten in place by the block output)
    bf16_t* bufB = bufA + SZ_A;                // intermediate of the current block
    bf16_t* s_wt = bufB + SZ_B;                // weights: L slots (resident) or one rotating slot
    // RESIDENT: the convs' biases and the follower's fragment + bias live in LDS too (BB_MISC_BYTES behind the weights).  Fetched
    // from global memory inside the tile loop they were loads BEHIND the next tile's prefetch in the in-order vector-memory queue:
    // their wait (s_waitcnt vmcnt(0)) held every conv's first MFMA until the whole prefetch had landed (round 5)
    float* s_misc = (float*)(s_wt + (RESIDENT ? 2 * NB * W_UNITS * 8 : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: row tests and row addresses on the scalar unit
    const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    u32x4 pin[IN_PT], pwt[W_PT];
    // byte offset of each of the thread's units from its tile's first halo pixel: the same for every tile (unused units of the last
    // round point at the first pixel, loaded and never committed)
    unsigned voff[IN_PT];
#pragma unroll
    for (int k = 0; k < IN_PT; ++k) {
        const int u = tid + k * 512;
        const int c8 = u % (C / 8), pix = u / (C / 8);
        voff[k] = u < IN_UNITS ? (unsigned)((((pix / R0W) * a.W + pix % R0W) * C + c8 * 8) * 2) : 0u;
    }
    auto issue_in = [&](int it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int gy0 = (tt / a.tiles_x) * TH - L, gx0 = (tt % a.tiles_x) * TW - L;
#ifndef TTUP_NO_FAST_PREFETCH
        if (gy0 >= 0 && gy0 + R0H <= a.H && gx0 >= 0 && gx0 + R0W <= a.W) {
            // the whole halo region lies inside the image (wave-uniform): a scalar base + the per-lane constants -- no coordinates, no
            // bounds tests, no 64-bit per-lane address arithmetic (round 5: the general form below is ~25 vector instructions per load,
            // issued while the matrix pipe has nothing to do)
            const char* base = (const char*)(a.x + ((size_t)(b * a.H + gy0) * a.W + gx0) * C);
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) pin[k] = *(const u32x4*)(base + opaque_u32(voff[k]));
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            const int c8 = u % (C / 8), pix = u / (C / 8);
            const int gy = gy0 + pix / R0W, gx = gx0 + pix % R0W;
            pin[k] = u32x4{0u, 0u, 0u, 0u};
            if (u < IN_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) pin[k] = *(const u32x4*)(a.x + ((size_t)(b * a.H + gy) * a.W + gx) * C + c8 * 8);
        }
    };
    auto load_wt = [&](int conv) {
#pragma unroll
        for (int k = 0; k < W_PT; ++k) { const int u = tid + k * 512; if (u < W_UNITS) pwt[k] = ((const u32x4*)a.w[conv])[u]; }
    };
    auto store_wt = [&](int slot) {
#pragma unroll
        for (int k = 0; k < W_PT; ++k) { const int u = tid + k * 512; if (u < W_UNITS) ((u32x4*)(s_wt + slot * W_UNITS * 8))[u] = pwt[k]; }
    };
    if (RESIDENT) {
        // the first tile and both convs' weights travel together: one round trip before the loop
        if (my_tiles > 0) issue_in(0);
        u32x4 pw2[2 * NB][W_PT];
#pragma unroll
        for (int cv = 0; cv < 2 * NB; ++cv)
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * 512; pw2[cv][k] = u32x4{0u, 0u, 0u, 0u}; if (u < W_UNITS) pw2[cv][k] = ((const u32x4*)a.w[cv])[u]; }
#pragma unroll
        for (int cv = 0; cv < 2 * NB; ++cv)
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int u = tid + k * 512; if (u < W_UNITS) ((u32x4*)(s_wt + cv * W_UNITS * 8))[u] = pw2[cv][k]; }
        // floats [0, C) bias of conv 0, [C, 2C) bias of conv 1, [2C, 2C+16) follower bias, then the follower's 64 x 16-byte fragment
        if (tid < C) { s_misc[tid] = a.bias[0][tid]; s_misc[C + tid] = a.bias[1][tid]; }
        if (a.yf) {
            if (tid < 16) s_misc[2 * C + tid] = a.bf[tid];
            if (tid >= 64 && tid < 128) ((u32x4*)(s_misc + 2 * C + 16))[tid - 64] = ((const u32x4*)a.wf)[tid - 64];
        }
    } else {
        if (!WGLOBAL && my_tiles > 0) load_wt(0);
        if (my_tiles > 0) issue_in(0);
    }

    if (my_tiles <= 0) return;          // (workgroup-uniform)
    if (RESIDENT) prefetch_arrived(pin);          // every path into the loop has the prefetch registers complete (see prefetch_arrived)
    for (int it = 0; it < my_tiles; ++it) {
        const int tl = xcd_tile(blockIdx.x + it * gridDim.x, a.total_tiles);
        const int b = tl / a.tiles_per_img, tt = tl % a.tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * TH, ox0 = (tt % a.tiles_x) * TW;
        if (C == 32) TTUP_STAMP_IT(2, it, 0);
        __syncthreads();                       // previous tile fully consumed (resident weights visible on the first pass)
        if (C == 32) TTUP_STAMP_IT(2, it, 1);
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int u = tid + k * 512;
            if (u < IN_UNITS) { const int c8 = u % (C / 8), pix = u / (C / 8); *(u32x4*)(bufA + bb_off<C>(pix, pix % R0W, c8)) = pin[k]; }
        }
        if (!WGLOBAL && !RESIDENT) store_wt(0);
        __syncthreads();
        if (it + 1 < my_tiles) issue_in(it + 1);
        if (C == 32) TTUP_STAMP_IT(2, it, 2);
        if (!WGLOBAL && !RESIDENT) load_wt(1);
        const bf16_t* w0 = WGLOBAL ? a.w[0] : s_wt;
        const bf16_t* w1 = WGLOBAL ? a.w[1] : (RESIDENT ? s_wt + W_UNITS * 8 : s_wt);
        if (NB == 1) {
            const float* bias0 = RESIDENT ? s_misc : a.bias[0];
            const float* bias1 = RESIDENT ? s_misc + C : a.bias[1];
            const bf16_t* wfl = RESIDENT ? (const bf16_t*)(s_misc + 2 * C + 16) : a.wf;
            const float* bfl = RESIDENT ? s_misc + 2 * C : a.bf;
            bb_conv<C, R0W, 0, R0H - 2, R0W - 2, false, 1, 0, false, R0W - 2, 0>(bufA, bufB, nullptr, w0, bias0, nullptr, oy0 - 1, ox0 - 1, a.H, a.W, b, wave, lane);
            if (C == 32) TTUP_STAMP_IT(2, it, 3);
            __syncthreads();
            if (C == 32) TTUP_STAMP_IT(2, it, 4);
            if (!WGLOBAL && !RESIDENT) { store_wt(0); __syncthreads(); if (it + 1 < my_tiles) load_wt(0); }
            // the next tile's input (requested before the first conv) is waited for HERE, in front of the second conv's stores
            if (RESIDENT) prefetch_arrived(pin);
            bb_conv<C, R0W - 2, 0, TH, TW, true, R0W, 2, true, 1, 0>(bufB, nullptr, bufA, w1, bias1, a.y, oy0, ox0, a.H, a.W, b, wave, lane, wfl, bfl, a.yf);
        } else {
            static_assert(NB == 1 || WGLOBAL, "two-block chains read their weights from global memory");
            bb_conv<C, R0W, 0, R0H - 2, R0W - 2, false, 1, 0, false, R0W - 2, 0>(bufA, bufB, nullptr, w0, a.bias[0], nullptr, oy0 - 3, ox0 - 3, a.H, a.W, b, wave, lane);
            __syncthreads();
            bb_conv<C, R0W - 2, 0, R0H - 4, R0W - 4, true, R0W, 2, false, R0W, 2>(bufB, bufA, bufA, w1, a.bias[1], nullptr, oy0 - 2, ox0 - 2, a.H, a.W, b, wave, lane);
            __syncthreads();
            bb_conv<C, R0W, 2, R0H - 6, R0W - 6, false, 1, 0, false, R0W - 6, 0>(bufA, bufB, nullptr, a.w[2], a.bias[2], nullptr, oy0 - 1, ox0 - 1, a.H, a.W, b, wave, lane);
            __syncthreads();
            bb_conv<C, R0W - 6, 0, TH, TW, true, R0W, 4, true, 1, 0>(bufB, nullptr, bufA, a.w[3], a.bias[3], a.y, oy0, ox0, a.H, a.W, b, wave, lane);
        }
    }
}

// One tile per workgroup, weights straight from L2 into registers (lowest register footprint: two workgroups per CU): the C=16 two-block
// chain with its fuse-sum / head epilogue configured at RUN time -- the fallback for term layouts other than HRNet's and the cross-check
// (TTUP_BB2_GENERIC=1) of c16_chain_kernel (csrc/chain16.h), which carries the forms the network uses and superseded this kernel's
// compiled-out variants in round 6.
template <int C, int TH, int TW>
__global__ __launch_bounds__(512, 4) void bb_chain2_kernel(BBArgs a) {       // 4 waves per SIMD = two workgroups per CU: at most 128 VGPRs
    constexpr int L = 4;
    constexpr int R0H = TH + 2 * L, R0W = TW + 2 * L;
    // row strides (pixels) of the two LDS buffers: ODD, so that 16 consecutive rows of one column fall on 16 different 16-byte
    // slots -- the strip groups of bb_conv are column groups (see there); 40 -> 41 and 38 -> 39 pixels cost 2 KB of LDS per workgroup
    constexpr int SA = (R0W & 1) ? R0W : R0W + 1;
    constexpr int SB = ((R0W - 2) & 1) ? R0W - 2 : R0W - 1;
    constexpr int SZ_A = R0H * SA * C;
    constexpr int SZ_B = (R0H - 2) * SB * C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* bufA = (bf16_t*)smem;
    bf16_t* bufB = bufA + SZ_A;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: row tests and row addresses on the scalar unit
    // 3-D grid (tile column, tile row, image): no division to find the tile
    // (XCD = linear workgroup id % 8 = blockIdx.x % 8 when the row has a multiple of 8 tiles: every XCD then takes a strip of
    // adjacent tile columns through all rows and images instead of every eighth column -- see xcd_tile)
#ifdef TTUP_NO_XCD_MAP
    const int bx = blockIdx.x;
#else
    const int bx = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
#endif
    const int b = blockIdx.z, tt = blockIdx.y * a.tiles_x + bx;
    const int oy0 = blockIdx.y * TH, ox0 = bx * TW;
#ifdef TTUP_TIMING
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    TTUP_STAMP(0);
    BBFrag16 fr;
    {
        // all of the thread's loads are issued before the first LDS store: ONE memory round trip for the tile, not one per unit.
        // A thread keeps one 16-byte column unit and walks rows (row lane rl, then every RL-th row): the global and the LDS
        // address of every further row are the first row's plus a constant -- no per-unit division, one bounds test per row.
        constexpr int CU = R0W * (C / 8);                 // 16-byte units per tile row
        constexpr int RL = 512 / CU;                      // row lanes
        constexpr int IN_PT = (R0H + RL - 1) / RL;
        static_assert(RL >= 1, "tile row wider than the workgroup");
        const int cu = tid % CU, rl = tid / CU;
        const int col = cu / (C / 8), c8 = cu % (C / 8);
        const int gx = ox0 - L + col, gyb = oy0 - L + rl;
        const bool col_ok = rl < RL && gx >= 0 && gx < a.W;
        const bf16_t* src = a.x + ((long long)(b * a.H + gyb) * a.W + gx) * C + c8 * 8;      // may point outside for halo rows / columns: only dereferenced when valid
        const long long row_step = (long long)RL * a.W * C;
        u32x4 v[IN_PT];
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int gy = gyb + k * RL;
            // branch-free: an invalid unit reads the tensor's first bytes and is zeroed afterwards (a branch around the load
            // would make every load wait for the one before it)
#ifdef TTUP_ABL_NOSTAGE
            const bool ok = false; (void)col_ok; (void)gy;
            const u32x4 t = u32x4{0u, 0u, 0u, 0u};
#else
            const bool ok = col_ok && rl + k * RL < R0H && gy >= 0 && gy < a.H;
            const u32x4 t = *(const u32x4*)(ok ? src + k * row_step : a.x);
#endif
            v[k] = u32x4{ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u};
        }
        bf16_t* dst = bufA + bb_off<C>(rl * SA + col, col, c8);
#pragma unroll
        for (int k = 0; k < IN_PT; ++k)
            if (rl < RL && rl + k * RL < R0H) *(u32x4*)(dst + k * RL * SA * C) = v[k];
    }
    const bf16x8 idm = bb_identity_frag(lane);
    if (C == 16) bb_load_frag16(fr, a.w[0], a.bias[0], lane);          // first conv's fragments: in flight across the barrier
    __syncthreads();
    TTUP_STAMP(1);
    const BBFrag16* pre = C == 16 ? &fr : nullptr;
    bb_conv<C, SA, 0, R0H - 2, R0W - 2, false, 1, 0, false, SB, 0>(bufA, bufB, nullptr, a.w[0], a.bias[0], nullptr, oy0 - 3, ox0 - 3, a.H, a.W, b, wave, lane,
                                                                            nullptr, nullptr, nullptr, nullptr, nullptr, pre);
    TTUP_STAMP(2);
    if (C == 16) bb_load_frag16(fr, a.w[1], a.bias[1], lane);          // next conv's fragments: requested BEFORE the barrier
    __syncthreads();
    TTUP_STAMP(3);
    bb_conv<C, SB, 0, R0H - 4, R0W - 4, true, SA, 2, false, SA, 2>(bufB, bufA, bufA, a.w[1], a.bias[1], nullptr, oy0 - 2, ox0 - 2, a.H, a.W, b, wave, lane,
                                                                            nullptr, nullptr, nullptr, nullptr, nullptr, pre, nullptr, idm);
    if (C == 16) bb_load_frag16(fr, a.w[2], a.bias[2], lane);
    __syncthreads();
    TTUP_STAMP(4);
    // The fuse-layer terms that the last conv's epilogue adds (1x1-conv'd lower branches at 1/2, 1/4, 1/8 resolution): the tile's
    // slices (12x16 + 6x8 + 3x4 pixels of 16 channels = 8 KB at most) are requested now, travel while conv3 runs, and are parked in
    // the tail of bufB that conv3's 26x34 output leaves free -- the epilogue then reads them from LDS instead of paying a memory
    // round trip per output row.
    constexpr int T_FREE = SZ_B - (TH + 2) * (TW + 2) * C;       // elements of bufB behind conv3's output
    static_assert(C != 16 || (TH % 8 == 0 && TW % 8 == 0), "term slices are aligned to the tile for 8-aligned tiles");
    bf16_t* s_terms = bufB + (TH + 2) * (TW + 2) * C;
    u32x4 treg = u32x4{0u, 0u, 0u, 0u};
    int tunit = -1;
    BBTermLds tlds;
    tlds.s_terms = s_terms;
    if (C == 16) {
        int base = 0;                 // in 16-byte units (two per pixel)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tlds.toff[k] = 0; tlds.tw[k] = 1;
            if (k < a.nsum) {
                const int sh = a.ssh[k], hk = TH >> sh, wk = TW >> sh;
                tlds.toff[k] = base * 8; tlds.tw[k] = wk;
                const int u = tid - base;
                if (u >= 0 && u < hk * wk * 2) {
                    const int px = u >> 1, ty = (oy0 >> sh) + px / wk, tx = (ox0 >> sh) + px % wk;
                    if (ty < (a.H >> sh) && tx < (a.W >> sh)) treg = *(const u32x4*)(a.st[k] + ((size_t)(b * (a.H >> sh) + ty) * (a.W >> sh) + tx) * 16 + (u & 1) * 8);
                    tunit = tid;
                }
                base += hk * wk * 2;
            }
        }
    }
    bb_conv<C, SA, 2, R0H - 6, R0W - 6, false, 1, 0, false, R0W - 6, 0>(bufA, bufB, nullptr, a.w[2], a.bias[2], nullptr, oy0 - 1, ox0 - 1, a.H, a.W, b, wave, lane,
                                                                            nullptr, nullptr, nullptr, nullptr, nullptr, pre);
    if (C == 16 && tunit >= 0) { static_assert(C != 16 || T_FREE * 2 >= ((TH >> 1) * (TW >> 1) + (TH >> 2) * (TW >> 2) + (TH >> 3) * (TW >> 3)) * 32, "bufB tail holds the term slices"); ((u32x4*)s_terms)[tunit] = treg; }
    if (C == 16) bb_load_frag16(fr, a.w[3], a.bias[3], lane);
    __syncthreads();
    TTUP_STAMP(5);
    BBBest best; best.v = -INFINITY; best.i = 0x7fffffffffffffffLL;
    bb_conv<C, R0W - 6, 0, TH, TW, true, SA, 4, true, 1, 0>(bufB, nullptr, bufA, a.w[3], a.bias[3], a.y, oy0, ox0, a.H, a.W, b, wave, lane,
                                                                   nullptr, nullptr, nullptr, &a, &best, pre, &tlds, idm);
#ifdef TTUP_TIMING_SPLIT
    TTUP_STAMP(6);
#endif
    if (C == 16 && a.heat) {
        // run-time form: lanes -> wave (shuffles) -> workgroup (through the now idle LDS)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_down(best.v, off, 64);
            const long long oi = __shfl_down(best.i, off, 64);
            if (bb_better(ov, oi, best.v, best.i)) { best.v = ov; best.i = oi; }
        }
        __syncthreads();                      // every wave is done with bufA / bufB
        float* sv = (float*)smem; long long* si = (long long*)(smem + 64);
        if (lane == 0) { sv[wave] = best.v; si[wave] = best.i; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < 8; ++k) if (bb_better(sv[k], si[k], best.v, best.i)) { best.v = sv[k]; best.i = si[k]; }
            a.pv[(size_t)b * a.tiles_per_img + tt] = best.v;
            a.pi[(size_t)b * a.tiles_per_img + tt] = best.i;
        }
    }
#ifndef TTUP_TIMING_SPLIT
    TTUP_STAMP(6);
#endif
#ifdef TTUP_TIMING
    if (tid == 0 && TTUP_BID < 8192) ttup_tbuf[TTUP_BID * 8 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;      // 100 MHz ticks for the same span
#endif
}

template <int C, int TH, int TW>
static int launch_bb2_t(const BBArgs& a, int batch, int h, int w, hipStream_t st) {
    constexpr int SA = ((TW + 8) & 1) ? TW + 8 : TW + 9, SB = ((TW + 6) & 1) ? TW + 6 : TW + 7;       // odd row strides, as in the kernel
    constexpr size_t SMEM = (size_t)((TH + 8) * SA + (TH + 6) * SB) * C * 2 + 64;       // + one argmax slot per wave
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(2 * SMEM <= 160 * 1024 || TH * TW > 24 * 32, "the 24x32 tile runs two workgroups per CU");
    if (int rc = ensure_max_lds((const void*)bb_chain2_kernel<C, TH, TW>, SMEM)) return rc;
    BBArgs k = a;
    k.H = h; k.W = w; k.tiles_x = cdiv(w, TW); k.tiles_per_img = k.tiles_x * cdiv(h, TH); k.total_tiles = k.tiles_per_img * batch;
    if (k.total_tiles == 0) return TTUP_OK;
    kernel_note("bb_chain2_kernel<%d, %d, %d>", C, TH, TW);
    hipLaunchKernelGGL((bb_chain2_kernel<C, TH, TW>), dim3(k.tiles_x, cdiv(h, TH), batch), dim3(512), SMEM, st, k);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

#include "chain16.h"

constexpr int BB_WT_SLOTS = 2;          // C=32: the weights of both convs of the block are LDS-resident
constexpr int BB_MISC_BYTES = (2 * 32 + 16) * 4 + 1024;      // ... and so are their biases and the follower's bias + fragment (bb_chain_kernel: s_misc)
template <int C, int NB, int TH, int TW>
static int launch_bb_t(const BBArgs& a, int batch, int h, int w, hipStream_t st) {
    constexpr int L = 2 * NB;
    constexpr int KSTEPS = (C == 16) ? 5 : 9, MT = C / 16;
    constexpr size_t SMEM = (size_t)((TH + 2 * L) * (TW + 2 * L) + (TH + 2 * L - 2) * (TW + 2 * L - 2)) * C * 2 +
                            (size_t)(C == 16 ? 0 : BB_WT_SLOTS) * KSTEPS * MT * 1024 + (C == 16 ? 0 : BB_MISC_BYTES);
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    if (int rc = ensure_max_lds((const void*)bb_chain_kernel<C, NB, TH, TW>, SMEM)) return rc;
    BBArgs k = a;
    k.H = h; k.W = w; k.tiles_x = cdiv(w, TW); k.tiles_per_img = k.tiles_x * cdiv(h, TH); k.total_tiles = k.tiles_per_img * batch;
    const int per_cu = (int)((160 * 1024) / SMEM) > 2 ? 2 : ((int)((160 * 1024) / SMEM) < 1 ? 1 : (int)((160 * 1024) / SMEM));
    const int grid = (C == 16 || k.total_tiles < 256 * per_cu) ? k.total_tiles : 256 * per_cu;      // C=16: one tile per workgroup
    if (grid == 0) return TTUP_OK;
    kernel_note("bb_chain_kernel<%d, %d, %d, %d>", C, NB, TH, TW);
    hipLaunchKernelGGL((bb_chain_kernel<C, NB, TH, TW>), dim3(grid), dim3(512), SMEM, st, k);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// tile of the C=16 two-block chain (multiples of 8: the fuse-term slices are aligned to the tile)
#ifndef TTUP_BB2_TH
#define TTUP_BB2_TH 24
#define TTUP_BB2_TW 32
#endif
constexpr int BB2_TH = TTUP_BB2_TH, BB2_TW = TTUP_BB2_TW;
int bb_chain_tiles_per_img(int h, int w) { return cdiv(w, BB2_TW) * cdiv(h, BB2_TH); }

int launch_bb_chain(const PackedConv* const* convs, int n_convs, const void* x, void* y, int batch, int h, int w,
                    const PackedConv* follow, void* y_follow, hipStream_t st, const BBSum* sum) {
    TTUP_REQUIRE(n_convs == 2 || n_convs == 4, TTUP_EINVAL, "bb_chain: 2 or 4 convs expected");
    const int c = convs[0]->cout;
    BBArgs a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y;
    a.wf = nullptr; a.bf = nullptr; a.yf = nullptr;
    a.nsum = 0; a.ysum = nullptr; a.heat = nullptr; a.hw = nullptr; a.hbias = 0.f; a.pv = nullptr; a.pi = nullptr;
    for (int k = 0; k < 3; ++k) { a.st[k] = nullptr; a.ssh[k] = 0; }
    if (sum) {
        TTUP_REQUIRE(c == 16 && n_convs == 4 && sum->n_terms >= 0 && sum->n_terms <= 3, TTUP_EINVAL, "bb_chain: the fused fuse-layer sum rides on the 16-channel two-block chain");
        TTUP_REQUIRE(sum->ysum || (sum->heat && sum->head_w && sum->pv && sum->pi), TTUP_EINVAL, "bb_chain: fused sum needs an output");
        a.nsum = sum->n_terms; a.ysum = (bf16_t*)sum->ysum;
        for (int k = 0; k < sum->n_terms; ++k) { a.st[k] = (const bf16_t*)sum->terms[k]; a.ssh[k] = sum->shifts[k]; }
        a.heat = sum->heat; a.hw = sum->head_w; a.hbias = sum->head_bias; a.pv = sum->pv; a.pi = sum->pi;
    }
    if (follow) {
        TTUP_REQUIRE(c == 32 && n_convs == 2 && follow->cout == 16 && follow->cin_total == 32 && follow->k == 1 && follow->ck == 32 && y_follow, TTUP_EINVAL,
                     "bb_chain: the fused follower is a 1x1 32->16 conv on a 32-channel block");
        a.wf = (const bf16_t*)follow->w_dev; a.bf = follow->bias_dev; a.yf = (bf16_t*)y_follow;
    }
    for (int i = 0; i < 4; ++i) { a.w[i] = nullptr; a.bias[i] = nullptr; }
    for (int i = 0; i < n_convs; ++i) {
        const PackedConv& p = *convs[i];
        TTUP_REQUIRE(p.cout == c && p.cin_total == c && p.k == 3 && p.stride == 1 && p.ck == (c == 16 ? 16 : 32), TTUP_EINVAL, "bb_chain: unexpected conv shape");
        a.w[i] = (const bf16_t*)p.w_dev; a.bias[i] = p.bias_dev;
    }
    // tile shapes tuned on MI355X: larger tiles amortise the per-tile overhead and waste fewer ragged 16-pixel MFMA groups
    if (c == 16 && n_convs == 4) {
        // the epilogue forms the network uses are compiled out in c16_chain_kernel (csrc/chain16.h); anything else -- other term layouts,
        // TTUP_BB2_GENERIC=1 (read once per process), other -DTTUP_BB2_TH/TW tiles -- takes the run-time form (bb_chain2_kernel)
        static const bool generic = getenv("TTUP_BB2_GENERIC") != nullptr;
#ifdef TTUP_ABL_EPI4          // timing build (wrong results): the plain chain for every launch
        return launch_c16_t<24, 32, 4>(a, batch, h, w, st);
#endif
        if constexpr (BB2_TH == 24 && BB2_TW == 32) {
            bool shifts_ok = true;          // the compiled-out forms assume term k at 1/2^(k+1) resolution (HRNet's fuse layers)
            for (int k = 0; k < a.nsum && k < 3; ++k) shifts_ok = shifts_ok && a.ssh[k] == k + 1;
            const bool sum_stored = !generic && shifts_ok && a.nsum >= 1 && a.nsum <= 3 && a.ysum && !a.heat;      // a.y (the pre-fuse tensor) optional
            const bool tail = !generic && shifts_ok && a.nsum == 3 && a.heat && !a.y && !a.ysum;
            const bool plain = !generic && a.nsum == 0 && !a.heat && !a.ysum && a.y;
            if (plain) return launch_c16_t<24, 32, 4>(a, batch, h, w, st);
            if (tail) return launch_c16_t<24, 32, 7>(a, batch, h, w, st);
            if (sum_stored && a.nsum == 1) return launch_c16_t<24, 32, 1>(a, batch, h, w, st);
            if (sum_stored && a.nsum == 2) return launch_c16_t<24, 32, 2>(a, batch, h, w, st);
            if (sum_stored && a.nsum == 3) return launch_c16_t<24, 32, 3>(a, batch, h, w, st);
        }
        return launch_bb2_t<16, BB2_TH, BB2_TW>(a, batch, h, w, st);
    }
    if (c == 16 && n_convs == 2) return launch_bb_t<16, 1, 8, 32>(a, batch, h, w, st);
    if (c == 32 && n_convs == 2) {
        // (a two-wave-group pipeline of this block, round 5: bit-identical and 1.5 % slower -- csrc/experiments/rejected_kernels.hip.inc)
        return launch_bb_t<32, 1, 22, 30>(a, batch, h, w, st);                   // conv regions 24x32 / 22x30
    }
    set_error("bb_chain: C=%d with %d convs unsupported", c, n_convs);
    return TTUP_EINVAL;
}

// ------------------------------------------------------------------ fp32 direct path (parity/debug)
struct ConvFArgs {
    const float* src0; const float* src1; const float* w; const float* bias; const float* residual; float* dst;
    int c0, c1, cout, ks, stride, H, W, OH, OW, relu;
    long long total;
};

__global__ void conv_direct_f32_kernel(ConvFArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int co = (int)(i % a.cout);
    long long p = i / a.cout;
    const int ox = (int)(p % a.OW); p /= a.OW;
    const int oy = (int)(p % a.OH);
    const int b = (int)(p / a.OH);
    const int pad = a.ks / 2, cin = a.c0 + a.c1;
    float acc = 0.f;
    for (int dy = 0; dy < a.ks; ++dy) {
        const int gy = oy * a.stride - pad + dy;
        if (gy < 0 || gy >= a.H) continue;
        for (int dx = 0; dx < a.ks; ++dx) {
            const int gx = ox * a.stride - pad + dx;
            if (gx < 0 || gx >= a.W) continue;
            const size_t pix = (size_t)(b * a.H + gy) * a.W + gx;
            const float* wt = a.w + (size_t)((dy * a.ks + dx) * cin) * a.cout + co;
            const float* s0 = a.src0 + pix * a.c0;
            for (int c = 0; c < a.c0; ++c) acc = fmaf(s0[c], wt[(size_t)c * a.cout], acc);
            if (a.c1) {
                const float* s1 = a.src1 + pix * a.c1;
                const float* wt1 = wt + (size_t)a.c0 * a.cout;
                for (int c = 0; c < a.c1; ++c) acc = fmaf(s1[c], wt1[(size_t)c * a.cout], acc);
            }
        }
    }
    acc += a.bias[co];
    if (a.residual) acc += a.residual[i];
    if (a.relu) acc = acc > 0.f ? acc : 0.f;
    a.dst[i] = acc;
}

// ------------------------------------------------------------------ host side: packing
void free_conv(PackedConv* p) {
    if (p->w_dev) (void)hipFree(p->w_dev);
    if (p->bias_dev) (void)hipFree(p->bias_dev);
    if (p->w3_dev) (void)hipFree(p->w3_dev);
    p->w_dev = nullptr; p->bias_dev = nullptr; p->w3_dev = nullptr;
}

int pack_conv(const FoldedConv& a, const FoldedConv* b, int cin_pad, int dtype, PackedConv* out) {
    const int k = a.k, taps = k * k, cout = a.cout;
    const int c0 = cin_pad > a.cin ? cin_pad : a.cin;
    const int c1 = b ? b->cin : 0;
    TTUP_REQUIRE(!b || (b->cout == cout && b->k == k && k == 1), TTUP_EINVAL, "two-source conv needs matching 1x1 convs");
    TTUP_REQUIRE(cout % 16 == 0 && cout <= 128, TTUP_EINVAL, "cout %d unsupported", cout);
    const int cin_total = c0 + c1;
    out->cout = cout; out->cin_total = cin_total; out->c0 = c0; out->k = k; out->stride = a.stride;
    std::vector<float> bias(cout);
    for (int i = 0; i < cout; ++i) bias[i] = a.bias[i] + (b ? b->bias[i] : 0.f);
    auto wval = [&](int co, int ci, int tap) -> float {
        if (ci < c0) return ci < a.cin ? a.w[((size_t)co * a.cin + ci) * taps + tap] : 0.f;
        return b->w[((size_t)co * b->cin + (ci - c0)) * taps + tap];
    };
    TTUP_HIP_CHECK(hipMalloc((void**)&out->bias_dev, cout * sizeof(float)));
    TTUP_HIP_CHECK(hipMemcpy(out->bias_dev, bias.data(), cout * sizeof(float), hipMemcpyHostToDevice));
    if (dtype == TTUP_DTYPE_F32) {
        std::vector<float> w((size_t)taps * cin_total * cout);
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < cin_total; ++ci)
                for (int co = 0; co < cout; ++co) w[((size_t)t * cin_total + ci) * cout + co] = wval(co, ci, t);
        out->w_bytes = w.size() * sizeof(float);
        out->ck = 0;
        TTUP_HIP_CHECK(hipMalloc(&out->w_dev, out->w_bytes));
        TTUP_HIP_CHECK(hipMemcpy(out->w_dev, w.data(), out->w_bytes, hipMemcpyHostToDevice));
        return pack_conv_x3(w, cout, cin_total, c0, k, a.stride, out);          // + the split-bf16 packing of the same weights (sets ck)
    }
    TTUP_REQUIRE(c0 % 16 == 0 && c1 % 32 == 0, TTUP_EINVAL, "channel counts %d+%d unsupported", c0, c1);
    int ck = (c0 % 32 == 0) ? 32 : 16;
    TTUP_REQUIRE(ck == 32 || (c1 == 0 && k == 3), TTUP_EINVAL, "16-channel chunks only for single-source 3x3");
    const int mt = cout / 16, ksteps = ck == 32 ? taps : (taps + 1) / 2, nchunk = cin_total / ck;
    std::vector<bf16_t> w((size_t)nchunk * ksteps * mt * 64 * 8);
    size_t idx = 0;
    for (int c = 0; c < nchunk; ++c)
        for (int s = 0; s < ksteps; ++s)
            for (int m = 0; m < mt; ++m)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    const int co = (i >> 2) * (4 * mt) + m * 4 + (i & 3);
                    for (int j = 0; j < 8; ++j) {
                        int tap, ci;
                        if (ck == 32) { tap = s; ci = c * 32 + 8 * g + j; }
                        else { tap = 2 * s + (g >> 1); ci = c * 16 + 8 * (g & 1) + j; }
                        w[idx++] = f32_to_bf16(tap < taps ? wval(co, ci, tap) : 0.f);
                    }
                }
    out->ck = ck;
    out->w_bytes = w.size() * sizeof(bf16_t);
    TTUP_HIP_CHECK(hipMalloc(&out->w_dev, out->w_bytes));
    TTUP_HIP_CHECK(hipMemcpy(out->w_dev, w.data(), out->w_bytes, hipMemcpyHostToDevice));
    return TTUP_OK;
}

// ------------------------------------------------------------------ launch
template <int CK, int COUT, int KS, int S, int TH, int TW, int NW, bool F11 = false>
static int launch_mfma(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    constexpr int MT = COUT / 16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int KSTEPS = (CK == 32) ? KS * KS : (KS * KS + 1) / 2;
    constexpr size_t SMEM = (size_t)(((IH * IW * CK + 7) & ~7) + KSTEPS * MT * 64 * 8) * 2;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    ConvKArgs a;
    a.src0 = (const bf16_t*)l.src0; a.src1 = (const bf16_t*)l.src1; a.wpack = (const bf16_t*)p.w_dev; a.bias = p.bias_dev;
    a.residual = (const bf16_t*)l.residual; a.dst = (bf16_t*)l.dst;
    a.c0 = p.c0; a.c1 = p.cin_total - p.c0; a.nchunk0 = p.c0 / CK; a.nchunk = p.cin_total / CK;
    a.H = l.h; a.W = l.w; a.OH = (l.h + S - 1) / S; a.OW = (l.w + S - 1) / S;
    a.tiles_x = cdiv(a.OW, TW);
    a.relu = l.relu;
    a.w11 = nullptr; a.bias11 = nullptr; a.dst11 = nullptr;
    a.res2 = (const bf16_t*)l.res2; a.res3 = (const bf16_t*)l.res3; a.sh3 = l.sh3;
    if (F11) {
        TTUP_REQUIRE(l.follow && l.follow->cout == 32 && l.follow->cin_total == 64 && l.follow->k == 1 && l.follow->ck == 32 && l.dst2, TTUP_EINVAL, "conv: bad fused 1x1 follower");
        a.w11 = (const bf16_t*)l.follow->w_dev; a.bias11 = l.follow->bias_dev; a.dst11 = (bf16_t*)l.dst2;
    }
    a.tiles_per_img = a.tiles_x * cdiv(a.OH, TH);
    a.total_tiles = a.tiles_per_img * l.batch;
    // XCD-aware tile order for the stride-2 convs (env TTUP_S2_XCD=0/1 overrides; the stride-1 full-resolution conv is 5-10 % slower with it)
    static const int s2_xcd = getenv("TTUP_S2_XCD") ? atoi(getenv("TTUP_S2_XCD")) : 0;
    a.xcd = (S == 2) ? s2_xcd : 0;
    // persistent grid: as many workgroups as can be resident (LDS-limited), each walks its share of the tiles
    const int per_cu = (int)((160 * 1024) / SMEM) > 4 ? 4 : ((int)((160 * 1024) / SMEM) < 1 ? 1 : (int)((160 * 1024) / SMEM));
    const int grid = a.total_tiles < 256 * per_cu ? a.total_tiles : 256 * per_cu;
    if (int rc = ensure_max_lds((const void*)conv_mfma_kernel<CK, COUT, KS, S, TH, TW, NW, F11>, SMEM)) return rc;
    if (grid == 0) return TTUP_OK;
    kernel_note("conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", CK, COUT, KS, S, TH, TW, NW, F11 ? "true" : "false");
    hipLaunchKernelGGL((conv_mfma_kernel<CK, COUT, KS, S, TH, TW, NW, F11>), dim3(grid), dim3(NW * 64), SMEM, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

template <int CK, int KS, int S, int TH, int TW>
static int dispatch_cout(const PackedConv& p, const ConvLaunch& l, hipStream_t st) {
    static const int nw = getenv("TTUP_CONV_WAVES") ? atoi(getenv("TTUP_CONV_WAVES")) : 8;
    // the stride-2 32 -> 64 conv alone is faster with four-wave workgroups (0.107 against 0.113 ms for its two launches, round 5: twice the
    // workgroups per CU behind its 52-KB staging); every other variant is 5-26 % slower that way
    static const bool nw_forced = getenv("TTUP_CONV_WAVES") != nullptr;
    if (!nw_forced && CK == 32 && S == 2 && p.cout == 64) return launch_mfma<CK, 64, KS, S, TH, TW, 4>(p, l, st);
    if (nw == 8) {
        switch (p.cout) {
            case 16: return launch_mfma<CK, 16, KS, S, TH, TW, 8>(p, l, st);
            case 32: return launch_mfma<CK, 32, KS, S, TH, TW, 8>(p, l, st);
            case 64: return launch_mfma<CK, 64, KS, S, TH, TW, 8>(p, l, st);
            case 128: return launch_mfma<CK, 128, KS, S, TH, TW, 8>(p, l, st);
        }
    }
    switch (p.cout) {
        case 16: return launch_mfma<CK, 16, KS, S, TH, TW, 4>(p, l, st);
        case 32: return launch_mfma<CK, 32, KS, S, TH, TW, 4>(p, l, st);
        case 64: return launch_mfma<CK, 64, KS, S, TH, TW, 4>(p, l, st);
        case 128: return launch_mfma<CK, 128, KS, S, TH, TW, 4>(p, l, st);
    }
    set_error("conv: cout %d unsupported", p.cout);
    return TTUP_EINVAL;
}

int launch_conv(const PackedConv& p, const ConvLaunch& l, int dtype, hipStream_t st) {
    if (dtype == TTUP_DTYPE_F32) {
        TTUP_REQUIRE(!l.res2 && !l.res3, TTUP_EINVAL, "conv: extra fuse-layer terms are a bf16-path fusion");
        static const bool direct = getenv("TTUP_F32_DIRECT") != nullptr;         // cross-checks: one thread per output, plain fp32 fma chain
        static const bool exact = getenv("TTUP_F32_EXACT") != nullptr;           // ... / exact fp32 products on the fp32 matrix pipe
        if (!direct && !exact && conv_x3_supported(p)) return launch_conv_x3(p, l, st);
        if (!direct && conv_f32_mfma_supported(p)) return launch_conv_f32_mfma(p, l, st);
        TTUP_REQUIRE(!l.n_active, TTUP_EINVAL, "conv: a device-side batch needs the matrix-pipe fp32 kernel");
        ConvFArgs a;
        a.src0 = (const float*)l.src0; a.src1 = (const float*)l.src1; a.w = (const float*)p.w_dev; a.bias = p.bias_dev;
        a.residual = (const float*)l.residual; a.dst = (float*)l.dst;
        a.c0 = p.c0; a.c1 = p.cin_total - p.c0; a.cout = p.cout; a.ks = p.k; a.stride = p.stride;
        a.H = l.h; a.W = l.w; a.OH = (l.h + p.stride - 1) / p.stride; a.OW = (l.w + p.stride - 1) / p.stride; a.relu = l.relu;
        a.total = (long long)l.batch * a.OH * a.OW * a.cout;
        const int threads = 256;
        const long long blocks = (a.total + threads - 1) / threads;
        kernel_note("conv_direct_f32_kernel");
        hipLaunchKernelGGL(conv_direct_f32_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, a);
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    TTUP_REQUIRE(!(l.res2 || l.res3) || (p.stride == 2 && !l.follow), TTUP_EINVAL, "conv: extra fuse-layer terms ride on the stride-2 chain convs only");
    if (l.follow) {
        TTUP_REQUIRE(p.k == 3 && p.stride == 1 && p.ck == 32 && p.cout == 64, TTUP_EINVAL, "conv: fused follower needs a 3x3 s1 conv with 64 outputs");
        return launch_mfma<32, 64, 3, 1, 8, 32, 8, true>(p, l, st);
    }
    if (p.k == 3 && p.stride == 1 && p.ck == 32 && p.cout == 64 && p.cin_total == 64 && p.c0 == 64 && !l.src1) return launch_conv64(p, l, st);
    if (l.pair) {
        const PackedConv& q = *l.pair;
        TTUP_REQUIRE(p.k == 3 && p.stride == 2 && p.ck == 16 && p.cin_total == 16 && p.cout == 32 && q.k == 3 && q.stride == 2 && q.ck == 16 &&
                     q.cin_total == 16 && q.cout == 16 && l.pair_dst && !l.src1, TTUP_EINVAL, "conv: the paired form is 3x3 s2 16 -> 32 with 3x3 s2 16 -> 16");
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        a.src0 = (const bf16_t*)l.src0; a.wpack = (const bf16_t*)p.w_dev; a.bias = p.bias_dev; a.residual = (const bf16_t*)l.residual; a.dst = (bf16_t*)l.dst;
        a.res2 = (const bf16_t*)l.res2; a.res3 = (const bf16_t*)l.res3; a.sh3 = l.sh3; a.relu = l.relu;
        a.wpack_b = (const bf16_t*)q.w_dev; a.bias_b = q.bias_dev; a.dst_b = (bf16_t*)l.pair_dst; a.relu_b = l.pair_relu;
        a.H = l.h; a.W = l.w; a.OH = (l.h + 1) / 2; a.OW = (l.w + 1) / 2;
        a.tiles_x = cdiv(a.OW, 32); a.tiles_per_img = a.tiles_x * cdiv(a.OH, 4); a.total_tiles = a.tiles_per_img * l.batch;
        constexpr size_t SMEM = (size_t)(((9 * 65 * 16 + 7) & ~7) + 5 * 3 * 64 * 8) * 2;
        if (int rc = ensure_max_lds((const void*)conv_s2_pair_kernel, SMEM)) return rc;
        const int grid = a.total_tiles < 256 * 4 ? a.total_tiles : 256 * 4;
        if (grid == 0) return TTUP_OK;
        kernel_note("conv_s2_pair_kernel");
        hipLaunchKernelGGL(conv_s2_pair_kernel, dim3(grid), dim3(512), SMEM, st, a);
        TTUP_LAUNCH_CHECK();
        return TTUP_OK;
    }
    TTUP_REQUIRE(!l.lin16 && !l.lin32, TTUP_EINVAL, "conv: linear 1x1 followers ride on the 64 -> 64 3x3 kernel only");
    if (p.k == 3 && p.stride == 1) return p.ck == 32 ? dispatch_cout<32, 3, 1, 8, 32>(p, l, st) : dispatch_cout<16, 3, 1, 8, 32>(p, l, st);
    if (p.k == 3 && p.stride == 2) return p.ck == 32 ? dispatch_cout<32, 3, 2, 4, 32>(p, l, st) : dispatch_cout<16, 3, 2, 4, 32>(p, l, st);
    if (p.k == 1 && p.stride == 1 && p.ck == 32) return dispatch_cout<32, 1, 1, 8, 32>(p, l, st);
    set_error("conv: k=%d stride=%d ck=%d unsupported", p.k, p.stride, p.ck);
    return TTUP_EINVAL;
}

// ------------------------------------------------------------------ pointwise kernels
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

struct UpsumArgs { const void* base; const void* t[3]; int shift[3]; int n; void* dst; int H, W, C; long long total; const int* n_active; long long per_sample; Roi roi; int batch; };

template <typename T>
__global__ void upsum_kernel(UpsumArgs a) {
    // every sample's whole tensor is walked; samples whose pruning flag is set produce only the op's cone region
    int nb = a.batch;
    if (a.n_active) nb = *a.n_active < nb ? *a.n_active : nb;
    const long long total = (long long)nb * a.H * a.W * a.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        long long p = i / a.C;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H);
        const int b = (int)(p / a.H);
        if (a.roi.flag) { const int f = a.roi.flag[b]; if (f != 0 && a.roi.outside(f, y, x)) continue; }
        float v = ld((const T*)a.base + i);
        for (int k = 0; k < a.n; ++k) {
            const int sh = a.shift[k], hh = a.H >> sh, ww = a.W >> sh;
            v += ld((const T*)a.t[k] + ((size_t)(b * hh + (y >> sh)) * ww + (x >> sh)) * a.C + c);
        }
        st((T*)a.dst + i, v > 0.f ? v : 0.f);
    }
}

// bf16 fast path: one lane = 8 channels (16 bytes) of one pixel; low-resolution terms are re-read by the 2^shift
// neighbours from L1/L2
__global__ __launch_bounds__(256) void upsum_bf16x8_kernel(UpsumArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over b*h*w*(C/8)
    if (i >= a.total) return;
    const int c8n = a.C >> 3;
    const int c8 = (int)(i % c8n);
    long long p = i / c8n;
    const int x = (int)(p % a.W); p /= a.W;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    float v[8];
    {
        const u32x4 r = *((const u32x4*)a.base + i);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = bf16_to_f32((bf16_t)(w[k] & 0xffff)); v[2 * k + 1] = bf16_to_f32((bf16_t)(w[k] >> 16)); }
    }
    for (int t = 0; t < a.n; ++t) {
        const int sh = a.shift[t], hh = a.H >> sh, ww = a.W >> sh;
        const u32x4 r = *((const u32x4*)a.t[t] + ((size_t)(b * hh + (y >> sh)) * ww + (x >> sh)) * c8n + c8);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32((bf16_t)(w[k] & 0xffff)); v[2 * k + 1] += bf16_to_f32((bf16_t)(w[k] >> 16)); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
    *((u32x4*)a.dst + i) = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}

int launch_upsum(const void* base, const void* const* terms, const int* shifts, int n_terms, void* dst,
                 int batch, int h, int w, int c, int dtype, hipStream_t stream, const int* n_active, const Roi* roi) {
    UpsumArgs a;
    a.base = base; a.n = n_terms; a.dst = dst; a.H = h; a.W = w; a.C = c; a.batch = batch;
    if (roi) a.roi = *roi;
    TTUP_REQUIRE(!a.roi.flag || dtype == TTUP_DTYPE_F32, TTUP_EINVAL, "upsum: output regions are an fp32-path feature");
    for (int k = 0; k < 3; ++k) { a.t[k] = k < n_terms ? terms[k] : nullptr; a.shift[k] = k < n_terms ? shifts[k] : 0; }
    a.total = (long long)batch * h * w * c;
    a.n_active = n_active; a.per_sample = (long long)h * w * c;
    if (a.total == 0) return TTUP_OK;
    long long blocks = (a.total + 255) / 256;
    if (n_active && blocks > 8192) blocks = 8192;             // grid-stride: the launch is sized for the largest batch
    TTUP_REQUIRE(!n_active || dtype == TTUP_DTYPE_F32, TTUP_EINVAL, "upsum: a device-side batch is an fp32-path feature");
    kernel_note(dtype == TTUP_DTYPE_F32 ? "upsum_kernel<float>" : c % 8 == 0 ? "upsum_bf16x8_kernel" : "upsum_kernel<bf16>");
    if (dtype == TTUP_DTYPE_F32) hipLaunchKernelGGL(upsum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (c % 8 == 0) {
        a.total /= 8;
        hipLaunchKernelGGL(upsum_bf16x8_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a);
    } else hipLaunchKernelGGL(upsum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int cin, int cpad, int hw, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over b*hw*cpad
    if (i >= total) return;
    const int c = (int)(i % cpad);
    const long long p = i / cpad;
    const int b = (int)(p / hw), pix = (int)(p % hw);
    st(dst + i, c < cin ? src[((size_t)b * cin + c) * hw + pix] : 0.f);
}

int launch_nchw_to_nhwc(const float* src, void* dst, int batch, int cin, int cpad, int h, int w, int dtype, hipStream_t stream) {
    const long long total = (long long)batch * h * w * cpad;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == TTUP_DTYPE_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, (float*)dst, cin, cpad, h * w, total);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, cin, cpad, h * w, total);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, float* dst, int c, int hw, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over b*c*hw (dst order)
    if (i >= total) return;
    const int pix = (int)(i % hw);
    const long long q = i / hw;
    const int ch = (int)(q % c), b = (int)(q / c);
    dst[i] = ld(src + ((size_t)b * hw + pix) * c + ch);
}

int launch_nhwc_to_nchw(const void* src, float* dst, int batch, int c, int h, int w, int dtype, hipStream_t stream) {
    const long long total = (long long)batch * h * w * c;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == TTUP_DTYPE_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)src, dst, c, h * w, total);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, dst, c, h * w, total);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// head: 1x1 conv 16 -> n_out selected output channels (+bias), fp32 NCHW (B, n_out, H, W) out
template <typename T, int CIN>
__global__ void head_kernel(const T* src, const float* w, const float* bias, int n_out, float* heat, long long hw, long long npix_max, const int* n_active, Roi roi, int W) {
    long long nb = npix_max / hw;
    if (n_active) nb = *n_active < nb ? *n_active : nb;
    const long long npix = nb * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        if (roi.flag) {
            const long long bq = i / hw, rem = i - bq * hw;
            const int y = (int)(rem / W), xq = (int)(rem % W);
            const int f = roi.flag[bq];
            if (f != 0 && roi.outside(f, y, xq)) continue;
        }
        float x[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) x[c] = ld(src + i * CIN + c);
        const long long b = i / hw, pix = i % hw;
        for (int k = 0; k < n_out; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc = fmaf(x[c], w[k * CIN + c], acc);
            heat[(b * n_out + k) * hw + pix] = acc + bias[k];
        }
    }
}

int launch_head(const void* src, const float* w_dev, const float* bias_dev, int n_out, float* heat, int batch, int h, int w, int cin, int dtype,
                hipStream_t stream, const int* n_active, const Roi* roi) {
    const Roi r = roi ? *roi : Roi();
    TTUP_REQUIRE(cin == 16, TTUP_EINVAL, "head expects 16 input channels, got %d", cin);
    const long long hw = (long long)h * w, npix = (long long)batch * hw;
    if (npix == 0) return TTUP_OK;
    long long blocks = (npix + 255) / 256;
    if (n_active && blocks > 8192) blocks = 8192;
    if (dtype == TTUP_DTYPE_F32) hipLaunchKernelGGL((head_kernel<float, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)src, w_dev, bias_dev, n_out, heat, hw, npix, n_active, r, w);
    else hipLaunchKernelGGL((head_kernel<bf16_t, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)src, w_dev, bias_dev, n_out, heat, hw, npix, n_active, r, w);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// ------------------------------------------------------------------ a1: uint8 frames -> normalised triples
// OpenCV INTER_LINEAR on uint8 (fixed point, 11-bit coefficients) + (x/255 - mean)/std, see
// oracle/glue_ref.py for the algorithm statement.  Parity of the resize is unpinned (cv2 absent offline).
struct PreArgs {
    const uint8_t* frames; void* out; int src_h, src_w, dst_h, dst_w, first_triple, n_triples, layout; long long total;
    double scale_x, scale_y;
    int nf;                // frames per sample: 3 (ball triples t,t+1,t+2) or 1 (table detector, single frame)
    const float* lut;      // [3][256]: (v/255 - mean[c]) / std[c] evaluated in fp64 on the host, rounded to fp32
    // crop mode (certified argmax): output sample j is the crop_h x crop_w window at (y0, x0) of triple `map`, records
    // {map, y0, x0, -} at crops[4*(crop0+j)], only the first *n_active samples are produced
    const int* crops; const int* n_active; int crop0, crop_h, crop_w;
};

__device__ __forceinline__ int cv_round(float v) { return (int)rintf(v); }

__device__ __forceinline__ void axis_tap_x(int d, double scale, int src_n, int& i0, int& i1, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src_n - 1) { f = 0.f; s = src_n - 1; }
    i0 = s; i1 = s + 1 < src_n ? s + 1 : src_n - 1;
    c1 = cv_round(f * 2048.f); c0 = cv_round((1.f - f) * 2048.f);
}
__device__ __forceinline__ void axis_tap_y(int d, double scale, int src_n, int& i0, int& i1, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    i0 = s < 0 ? 0 : (s > src_n - 1 ? src_n - 1 : s);
    i1 = s + 1 < 0 ? 0 : (s + 1 > src_n - 1 ? src_n - 1 : s + 1);
    c1 = cv_round(f * 2048.f); c0 = cv_round((1.f - f) * 2048.f);
}

template <typename T>
__global__ void preprocess_kernel(PreArgs a) {
    // one thread per (sample, y, x): produces the 3*nf channels of that pixel.  Whole frames run on a 3-D grid (column block,
    // row, sample) -- no index division, which used to be most of this kernel's instructions; crop windows keep a linear index
    int x, y, t;
    size_t opix;                        // output pixel index (sample-major)
    if (a.crops) {
        const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (unsigned)a.total) return;
        const int cx = (int)(i % (unsigned)a.crop_w);
        const unsigned p = i / (unsigned)a.crop_w;
        const int cy = (int)(p % (unsigned)a.crop_h);
        const int j = (int)(p / (unsigned)a.crop_h);
        if (j >= *a.n_active) return;
        const int* rec = a.crops + 4 * (a.crop0 + j);
        t = rec[0]; y = rec[1] + cy; x = rec[2] + cx;
        opix = ((size_t)j * a.crop_h + cy) * a.crop_w + cx;
    } else {
        x = blockIdx.x * blockDim.x + threadIdx.x; y = blockIdx.y; t = blockIdx.z;
        if (x >= a.dst_w) return;
        opix = ((size_t)t * a.dst_h + y) * a.dst_w + x;
    }
    const bool same = a.src_h == a.dst_h && a.src_w == a.dst_w;
    int x0 = x, x1 = x, a0 = 2048, a1 = 0, y0 = y, y1 = y, b0 = 2048, b1 = 0;
    if (!same) {
        axis_tap_x(x, a.scale_x, a.src_w, x0, x1, a0, a1);
        axis_tap_y(y, a.scale_y, a.src_h, y0, y1, b0, b1);
    }
    float vals[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) vals[k] = 0.f;
    // the three channel bytes of a source pixel come in one unaligned 4-byte load (the fourth byte is the next pixel's
    // first channel); only the very last pixel of the clip falls back to byte loads so that nothing is read past the end
    typedef unsigned int __attribute__((aligned(1))) u32_unaligned;
    const size_t frame_bytes = (size_t)a.src_h * a.src_w * 3;
    const uint8_t* clip_last = a.frames + (size_t)(a.first_triple + a.n_triples + a.nf - 1) * frame_bytes - 4;
    auto load3 = [&](const uint8_t* q) -> unsigned {
        if (q <= clip_last) return *(const u32_unaligned*)q;
        return (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16);
    };
    for (int f = 0; f < a.nf; ++f) {
        const uint8_t* img = a.frames + (size_t)(a.first_triple + t + f) * frame_bytes;
        if (same) {
            const unsigned w = load3(img + ((size_t)y * a.src_w + x) * 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) vals[f * 3 + c] = a.lut[c * 256 + ((w >> (8 * c)) & 255)];
            continue;
        }
        // a tap with weight 0 (equal widths: every second x tap; integer row positions) is not loaded: 0 * v adds nothing
        const unsigned p00 = load3(img + ((size_t)y0 * a.src_w + x0) * 3);
        const unsigned p01 = a1 ? load3(img + ((size_t)y0 * a.src_w + x1) * 3) : 0u;
        const unsigned p10 = b1 ? load3(img + ((size_t)y1 * a.src_w + x0) * 3) : 0u;
        const unsigned p11 = (a1 && b1) ? load3(img + ((size_t)y1 * a.src_w + x1) * 3) : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sh = 8 * c;
            const int top = (int)((p00 >> sh) & 255) * a0 + (int)((p01 >> sh) & 255) * a1;
            const int bot = (int)((p10 >> sh) & 255) * a0 + (int)((p11 >> sh) & 255) * a1;
            int v = (((b0 * (top >> 4)) >> 16) + ((b1 * (bot >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            vals[f * 3 + c] = a.lut[c * 256 + v];
        }
    }
    const size_t hw = (size_t)a.dst_h * a.dst_w, pix = (size_t)y * a.dst_w + x;
    if (a.layout == TTUP_LAYOUT_NHWC4_FRAME) {          // one 4-channel bf16 record per frame pixel (stem frames mode)
        *(u32x2*)((bf16_t*)a.out + opix * 4) = u32x2{pack2(vals[0], vals[1]), pack2(vals[2], 0.f)};
        return;
    }
    if (a.layout == TTUP_LAYOUT_NCHW_F32) {
        float* o = (float*)a.out + (size_t)t * 3 * a.nf * hw + pix;
        for (int c = 0; c < 3 * a.nf; ++c) o[c * hw] = vals[c];
    } else {
        T* o = (T*)a.out + opix * 16;
        if (sizeof(T) == 2) {
            u32x4* o4 = (u32x4*)o;
            o4[0] = u32x4{pack2(vals[0], vals[1]), pack2(vals[2], vals[3]), pack2(vals[4], vals[5]), pack2(vals[6], vals[7])};
            o4[1] = u32x4{pack2(vals[8], 0.f), 0u, 0u, 0u};
        } else {
            for (int c = 0; c < 16; ++c) st(o + c, c < 9 ? vals[c] : 0.f);
        }
    }
}

// Frame records (one 4-channel bf16 record per pixel of ONE frame: the production input of the stem) when source and network
// width are equal, as for 1280x720 frames at 1280x704 -- the horizontal taps are (2048, 0), only rows are interpolated.  Four
// pixels of four rows per thread: the 12 source bytes of a row are three aligned words, the normalisation table sits in LDS (the general
// kernel's three dependent table loads per pixel from memory were what it waited for), two 16-byte stores.  Same integer arithmetic
// per pixel as preprocess_kernel: bit-identical records.
constexpr int PRE4_ROWS = 4;          // output rows per workgroup: the table load and its barrier are paid once for 4096 pixels
__global__ __launch_bounds__(256) void preprocess_frames4_kernel(PreArgs a) {
    __shared__ float s_lut[768];
    const int x = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4, yb = (int)blockIdx.y * PRE4_ROWS, t = blockIdx.z;
    const bool live = x < a.dst_w;
    const uint8_t* img = a.frames + (size_t)(a.first_triple + t) * a.src_h * a.src_w * 3;
    int b0[PRE4_ROWS], b1[PRE4_ROWS];
    u32x4 r0[PRE4_ROWS], r1[PRE4_ROWS];
    // all loads of the workgroup's rows first (one memory round trip), the table while they travel
#pragma unroll
    for (int r = 0; r < PRE4_ROWS; ++r) {
        const int y = yb + r;
        int y0 = y, y1 = y;
        b0[r] = 2048; b1[r] = 0;
        r0[r] = u32x4{0u, 0u, 0u, 0u}; r1[r] = u32x4{0u, 0u, 0u, 0u};
        if (y >= a.dst_h) continue;
        if (a.src_h != a.dst_h) axis_tap_y(y, a.scale_y, a.src_h, y0, y1, b0[r], b1[r]);
        if (live) {
            const unsigned* p0 = (const unsigned*)(img + ((size_t)y0 * a.src_w + x) * 3);
            r0[r] = u32x4{p0[0], p0[1], p0[2], 0u};
            if (b1[r]) {                                     // wave-uniform (a row property): a tap with weight 0 is not loaded
                const unsigned* p1 = (const unsigned*)(img + ((size_t)y1 * a.src_w + x) * 3);
                r1[r] = u32x4{p1[0], p1[1], p1[2], 0u};
            }
        }
    }
    for (int k = threadIdx.x; k < 768; k += 256) s_lut[k] = a.lut[k];
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int r = 0; r < PRE4_ROWS; ++r) {
        const int y = yb + r;
        if (y >= a.dst_h) break;
        const unsigned w0[3] = {r0[r].x, r0[r].y, r0[r].z}, w1[3] = {r1[r].x, r1[r].y, r1[r].z};
        unsigned rec[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int i = 3 * j + c;
                const int top = (int)((w0[i >> 2] >> (8 * (i & 3))) & 255) * 2048;
                const int bot = (int)((w1[i >> 2] >> (8 * (i & 3))) & 255) * 2048;
                int q = (((b0[r] * (top >> 4)) >> 16) + ((b1[r] * (bot >> 4)) >> 16) + 2) >> 2;
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                v[c] = s_lut[c * 256 + q];
            }
            rec[2 * j] = pack2(v[0], v[1]); rec[2 * j + 1] = pack2(v[2], 0.f);
        }
        u32x4* o = (u32x4*)((bf16_t*)a.out + (((size_t)t * a.dst_h + y) * a.dst_w + x) * 4);
        o[0] = u32x4{rec[0], rec[1], rec[2], rec[3]};
        o[1] = u32x4{rec[4], rec[5], rec[6], rec[7]};
    }
}

int launch_preprocess(const uint8_t* frames, int n_frames, int src_h, int src_w, int dst_h, int dst_w,
                      void* out, int out_layout, int dtype, int first_triple, int n_triples, int frames_per_sample, hipStream_t stream) {
    TTUP_REQUIRE(frames_per_sample == 1 || frames_per_sample == 3, TTUP_EINVAL, "frames_per_sample must be 1 or 3");
    TTUP_REQUIRE(first_triple >= 0 && first_triple + n_triples + frames_per_sample - 1 <= n_frames, TTUP_EINVAL, "sample range outside the clip");
    PreArgs a;
    a.frames = frames; a.out = out; a.src_h = src_h; a.src_w = src_w; a.dst_h = dst_h; a.dst_w = dst_w;
    a.first_triple = first_triple; a.n_triples = n_triples; a.layout = out_layout; a.nf = frames_per_sample;
    a.total = (long long)n_triples * dst_h * dst_w;
    a.scale_x = (double)src_w / dst_w; a.scale_y = (double)src_h / dst_h;
    a.crops = nullptr; a.n_active = nullptr; a.crop0 = 0; a.crop_h = 0; a.crop_w = 0;
    if (a.total == 0) return TTUP_OK;
    if (int rc = device_normalise_lut(&a.lut)) return rc;       // one table per device
    TTUP_REQUIRE(dst_h <= 65535 && n_triples <= 65535, TTUP_EINVAL, "preprocess: grid limit (rows, samples <= 65535)");
    const dim3 grid((unsigned)cdiv(dst_w, 256), (unsigned)dst_h, (unsigned)n_triples);
    TTUP_REQUIRE(out_layout != TTUP_LAYOUT_NHWC4_FRAME || (frames_per_sample == 1 && dtype == TTUP_DTYPE_BF16), TTUP_EINVAL, "per-frame records are bf16, one frame per sample");
    static const bool no_fast = getenv("TTUP_NO_PRE4") != nullptr;
    if (out_layout == TTUP_LAYOUT_NHWC4_FRAME && src_w == dst_w && dst_w % 4 == 0 && ((size_t)frames & 3) == 0 && !no_fast)      // aligned 12-byte row loads
        hipLaunchKernelGGL(preprocess_frames4_kernel, dim3((unsigned)cdiv(dst_w, 1024), (unsigned)cdiv(dst_h, PRE4_ROWS), (unsigned)n_triples), dim3(256), 0, stream, a);
    else if ((dtype == TTUP_DTYPE_F32 || out_layout == TTUP_LAYOUT_NCHW_F32) && out_layout != TTUP_LAYOUT_NHWC4_FRAME)
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, dim3(256), 0, stream, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// crop mode: fp32 NHWC16 windows of the pre-processed triples, chosen on the device (csrc/certify.hip)
int launch_preprocess_crops(const uint8_t* frames, int n_frames, int src_h, int src_w, int dst_h, int dst_w, float* out,
                            const int* crops_dev, int crop0, const int* n_active_dev, int max_crops, int crop_h, int crop_w,
                            int frames_per_sample, hipStream_t stream) {
    PreArgs a;
    a.frames = frames; a.out = out; a.src_h = src_h; a.src_w = src_w; a.dst_h = dst_h; a.dst_w = dst_w;
    a.first_triple = 0; a.n_triples = n_frames - frames_per_sample + 1; a.layout = TTUP_LAYOUT_NHWC16; a.nf = frames_per_sample;
    a.total = (long long)max_crops * crop_h * crop_w;
    a.scale_x = (double)src_w / dst_w; a.scale_y = (double)src_h / dst_h;
    a.crops = crops_dev; a.n_active = n_active_dev; a.crop0 = crop0; a.crop_h = crop_h; a.crop_w = crop_w;
    if (a.total == 0) return TTUP_OK;
    if (int rc = device_normalise_lut(&a.lut)) return rc;
    TTUP_REQUIRE(a.total < (1ll << 32), TTUP_EINVAL, "preprocess crops: more than 2^32 crop pixels");
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

}  // namespace ttup

#ifdef TTUP_TIMING
extern "C" int ttup_debug_read_timing(unsigned long long* out_host, int n_words) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ttup::ttup_tbuf), (size_t)n_words * sizeof(unsigned long long));
}
extern "C" int ttup_debug_read_timing_it(unsigned long long* out_host, int n_words) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ttup::ttup_tbuf_it), (size_t)n_words * sizeof(unsigned long long));
}
#endif
