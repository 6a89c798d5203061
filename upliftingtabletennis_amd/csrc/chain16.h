// The 16-channel two-block chain of the full-resolution branch, round-6 form: four 3x3 convs 16 -> 16 (two BasicBlocks,
// balldetection/models/wasb.py:48-64) per 24x32 output tile with every intermediate in LDS, and the fuse-layer sum that consumes the
// branch (wasb.py:227-245) -- or, in stage 4, the 1x1 head + argmax partial (wasb.py:484, 606) -- in the last conv's epilogue.
// Included by conv.hip (uses its BBArgs / BBFrag16 / pack2 / relu_pk / bb_key helpers).  Same tiling, LDS images, k-step order and
// rounding points as bb_chain2_kernel (csrc/conv.hip: the run-time-epilogue fallback and cross-check); what changed, from the round-5
// counters (VALU 4.3-5.7 per MFMA, SQ_LDS_BANK_CONFLICT 17 % of the LDS cycles, phase stamps: staging 4.4 k + conv4 6.3 k of 24 k cycles):
//   * conv1-3 store TWO 16-pixel groups per ds_write_b128: v_permlane16_swap turns (group A, group B) x (couts 4g..4g+3) into
//     (pixel of A | pixel of B) x (8-channel chunk), 8 consecutive lanes then cover 8 different 16-byte bank groups -- the 2-way conflict
//     of the 8-byte stores at a 32-byte pixel pitch is gone and the store count halves; the ragged strips are walked as PAIRS of column
//     groups (one strip column, rows 0..RHO/2-1 | RHO/2..RHO-1) and stored the same way;
//   * the fuse-layer terms are added ON THE MATRIX PIPE: one v_mfma with an identity A operand and the conv output as its C operand
//     sums two terms (K = 2 x 16 channels) -- 1 ds_read_b128 + 1 MFMA instead of 2 x (ds_read_b64 + 8 vector instructions) per group;
//   * the head's argmax partial keeps (value, group id) per lane with ONE compare per group (a lane meets its pixels in index order);
//     NaN / -inf heat values take a slow path that applies torch.argmax's rules to the six values the lane still holds;
//   * tiles whose halo region lies inside the image (89 %) load it with scalar row bases + one per-lane offset and no selects;
//   * the first conv's weight fragments and the identity fragment (a 1-KB table instead of ~35 instructions) are requested before
//     the tile, so that they arrive with it instead of costing a second memory round trip behind the first barrier.
#pragma once

// -DTTUP_TIMING -DTTUP_TIMING_C16W (tools/build_ablate.sh TIMING,TIMING_C16W; tools/c16_wave_timing.py): s_memtime stamps of EVERY wave of the
// first 512 workgroups at the phase boundaries, kept in scalar registers until the kernel ends
#if defined(TTUP_TIMING) && defined(TTUP_TIMING_C16W)
#define C16_WSTAMP(k) do { wst[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C16_WSTAMP(k) do { } while (0)
#endif

struct C16IdmTab {
    unsigned short v[64 * 8];
    constexpr C16IdmTab() : v() {
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) v[l * 8 + j] = ((l & 15) == ((l >> 4) & 1) * 8 + j) ? 0x3F80 : 0;
    }
};
// lane (n, g): row n of the 16 x 16 identity, columns (g & 1) * 8 .. + 7 -- as an A fragment it adds K-slot (g & 1) * 8 + j of lane group g to cout n
__device__ const C16IdmTab c16_idm_tab = C16IdmTab();

constexpr int C16_PB = 32;                  // bytes per pixel record (16 bf16 channels)
// L2 prefetch distance in workgroups (0 = off): see c16_chain_kernel.  Measured (two boxes, ms per launch, sum / stored-output forms):
// off 0.7184-0.7440; 512: -1.4 %; 256: -1.3 ... -2.3 % (-1.6 ... -3.2 % on the stored-output forms); 192 like 256; 128: +0.4 %; 64: +2.7 %
#ifndef TTUP_C16_PREFETCH
#define TTUP_C16_PREFETCH 256
#endif
// work split of conv1 / conv2 / conv3 of the 24x32 tile (regions 30x38, 28x36, 26x34): rows for waves 0-3 | 4-5 | 6-7, strip pairs likewise
// (measured against equal bands of four rows with the strip pairs dealt to the last wave first: -1.2 ... -1.7 % on the launches with
// stored outputs, +-0 on the stage-4 form; two neighbouring splits -- conv2 at 4 | 3 | 3 rows, or 5-row bands for the older waves in
// all three convs -- are within 0.5 % of this one)
#define C16_SPLIT1 4, 4, 3, 1, 0, 1
#define C16_SPLIT2 5, 2, 2, 0, 1, 1
#define C16_SPLIT3 4, 3, 2, 0, 0, 1
// byte offset of 16-byte chunk c8 of buffer column x inside a row of pixel records (the swizzle of bb_off<16>)
__device__ __forceinline__ int c16_col(int x, int c8) { return x * C16_PB + ((c8 ^ ((x >> 2) & 1)) << 4); }

struct C16Pair { unsigned a, b; };
// v_permlane16_swap: rows (16 lanes) 1 and 3 of x trade places with rows 0 and 2 of y
// (the results are pinned by an empty asm: hipcc otherwise SINKS the swap into a divergent `if (live) store` block,
// where the partner lane of a live lane may be masked off and hands over nothing)
__device__ __forceinline__ C16Pair c16_swap(unsigned x, unsigned y) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    unsigned a = r[0], b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
    return C16Pair{a, b};
}
// two 16-pixel groups' packed outputs (lane (n, g): couts 4g..4g+3 of pixel n of group A in a0|a1, of group B in b0|b1) -> the
// 16-byte chunk g >> 1 of pixel n of group (g & 1)
__device__ __forceinline__ u32x4 c16_pair_chunk(unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
    const C16Pair s0 = c16_swap(a0, b0), s1 = c16_swap(a1, b1);
    return u32x4{s0.a, s1.a, s0.b, s1.b};
}

// One 3x3 conv of the chain with its output in LDS (conv1-3).  Byte addresses; strides in pixels.  Region RHO x RWO, RWO = 32 + RX.
// Work split (SPLIT = RA, RY1, RY2 rows and PA, PY1, PY2 strip pairs for waves 0-3 | 4-5 | 6-7): the two waves of this workgroup on a
// SIMD are waves w and w + 4, and the SIMD's arbiter serves the OLDER one first (per-wave stamps, round 6: the same eight 16-pixel
// groups take waves 0-3 3.0-3.1 k cycles and waves 4-5 3.8-3.9 k) -- so the older waves get more of the band and the younger ones the
// strip pairs, instead of equal bands with the strip pairs dealt to whoever comes first.
template <int RWI, int IOFF, int RHO, int RWO, bool SECOND, int RWR, int ROFF, int ORW, int OOFF, int RA, int RY1, int RY2, int PA, int PY1, int PY2>
__device__ __forceinline__ void c16_conv_lds(const char* s_in, char* s_out, const char* s_res, const BBFrag16& fr, bf16x8 idm,
                                             int gy0, int gx0, int H, int W, int wave, int lane) {
    constexpr int RS = RWI * C16_PB, RSR = RWR * C16_PB, OS = ORW * C16_PB;
    constexpr int RB = RA > RY1 ? (RA > RY2 ? RA : RY2) : (RY1 > RY2 ? RY1 : RY2);          // most rows a wave walks
    constexpr int RX = RWO - 32;
    static_assert(RX > 0 && RX < 16 && (RHO & 1) == 0, "two full 16-pixel groups + a ragged strip, even row count");
    static_assert(4 * RA + 2 * RY1 + 2 * RY2 == RHO && 4 * PA + 2 * PY1 + 2 * PY2 == RX, "the split covers the band and the strip");
    // (wave-uniform) this wave's rows [yb, yb + rows) and strip pairs [p0, p0 + np)
    const int rows = wave < 4 ? RA : (wave < 6 ? RY1 : RY2);
    const int yb = wave < 4 ? wave * RA : (wave < 6 ? 4 * RA + (wave - 4) * RY1 : 4 * RA + 2 * RY1 + (wave - 6) * RY2);
    const int np = wave < 4 ? PA : (wave < 6 ? PY1 : PY2);
    const int p0 = wave < 4 ? wave * PA : (wave < 6 ? 4 * PA + (wave - 4) * PY1 : 4 * PA + 2 * PY1 + (wave - 6) * PY2);
    const int n = lane & 15, g = lane >> 4, h = g >> 1, c8 = g & 1;
    bf16x8 af[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) af[s] = fr.af[s];
    if (SECOND && g >= 2) af[4] = idm;          // the block input rides in the unused tenth tap (exact: bf16 x 1.0 into the fp32 sum)
    const f32x4 bias = fr.bias;
#ifdef TTUP_ABL_NOPAD
    const bool interior = true;
#else
    const bool interior = gy0 >= 0 && gy0 + RHO <= H && gx0 >= 0 && gx0 + RWO <= W;
#endif
    // ---- the band: rows yb .. yb+rows-1, column groups 0 and 1
    if (rows > 0) {
        const char* rowb = s_in + ((yb + IOFF) * RWI) * C16_PB;
        const char* pA = rowb + c16_col(IOFF + n + h, c8);                     // steps 0-2: row r+dy, column x | x+1
        const char* pC = rowb + h * RS + c16_col(IOFF + n + 2, c8);            // step 3: column x+2 of rows r | r+1
        const char* pD = rowb + 2 * RS + c16_col(IOFF + n + 2, c8);            // step 4: (r+2, x+2) | block input (second conv) / same pixel, zero weights
        int dstep = RS;
        if (SECOND && h) { pD = s_res + ((yb + ROFF) * RWR) * C16_PB + c16_col(ROFF + n, c8); dstep = RSR; }
        char* so = s_out + ((yb + OOFF) * ORW) * C16_PB + c16_col(OOFF + (g & 1) * 16 + n, g >> 1);
        const bool cin0 = (unsigned)(gx0 + n) < (unsigned)W, cin1 = (unsigned)(gx0 + 16 + n) < (unsigned)W;
        bf16x8 fa[2][RB + 2];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) { fa[xt][0] = *(const bf16x8*)(pA + xt * 512); fa[xt][1] = *(const bf16x8*)(pA + RS + xt * 512); }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int y = yb + r;
            if (r >= rows) break;
            bf16x8 f3[2], f4[2];
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                fa[xt][r + 2] = *(const bf16x8*)(pA + (r + 2) * RS + xt * 512);
                f3[xt] = *(const bf16x8*)(pC + r * RS + xt * 512);
                f4[xt] = *(const bf16x8*)(pD + r * dstep + xt * 512);
            }
            f32x4 acc[2];
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], fa[xt][r], bias, 0, 0, 0);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], fa[xt][r + 1], acc[xt], 0, 0, 0);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], fa[xt][r + 2], acc[xt], 0, 0, 0);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3], f3[xt], acc[xt], 0, 0, 0);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4], f4[xt], acc[xt], 0, 0, 0);
            unsigned p00 = relu_pk(pack2(acc[0][0], acc[0][1])), p01 = relu_pk(pack2(acc[0][2], acc[0][3]));
            unsigned p10 = relu_pk(pack2(acc[1][0], acc[1][1])), p11 = relu_pk(pack2(acc[1][2], acc[1][3]));
            if (!interior) {          // zero padding of the next conv: outputs outside the image are 0 (border tiles only, wave-uniform test)
                const bool row_in = (unsigned)(gy0 + y) < (unsigned)H;
                const bool in0 = row_in && cin0, in1 = row_in && cin1;
                p00 = in0 ? p00 : 0u; p01 = in0 ? p01 : 0u; p10 = in1 ? p10 : 0u; p11 = in1 ? p11 : 0u;
            }
            *(u32x4*)(so + r * OS) = c16_pair_chunk(p00, p01, p10, p11);
        }
    }
    // ---- the ragged strip (columns 32 .. 32+RX-1): pairs of COLUMN groups -- lane n = row n (group A) and row RHO/2 + n (group B) of one
    // strip column.  On the odd row strides of the chain's buffers 16 consecutive rows of a column fall on 16 different 16-byte slots,
    // like the 16 consecutive pixels of a band group (bb_conv's strip groups, round 4); same k-step order and operands per output pixel.
#ifndef TTUP_ABL_NOSTRIPWORK
    {
        constexpr int RPG = RHO / 2;
        static_assert(RPG <= 16, "a strip column is two 16-lane groups");
        const int nr = n < RPG ? n : RPG - 1;              // idle lanes recompute a neighbour's pixel (same addresses: broadcast) and store nothing
        auto pair = [&](int p) __attribute__((always_inline)) {
            const int col = 32 + p;
            const char* b0 = s_in + ((nr + IOFF) * RWI + IOFF + col) * C16_PB;
            const int sw2 = ((c8 ^ (((col + 2 + IOFF) >> 2) & 1)) << 4);
            const char* a0 = b0 + h * C16_PB + ((c8 ^ (((col + h + IOFF) >> 2) & 1)) << 4);      // steps 0-2: rows row + dy, column col | col+1
            const char* a3 = b0 + h * RS + 2 * C16_PB + sw2;                                    // step 3: column col+2 of rows row | row+1
            const char* a4 = b0 + 2 * RS + 2 * C16_PB + sw2;                                    // step 4: (row+2, col+2) | block input / pad
            int a4step = RPG * RS;
            if (SECOND && h) { a4 = s_res + ((nr + ROFF) * RWR + ROFF + col) * C16_PB + ((c8 ^ (((col + ROFF) >> 2) & 1)) << 4); a4step = RPG * RSR; }
            f32x4 acc[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], *(const bf16x8*)(a0 + q * RPG * RS), bias, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], *(const bf16x8*)(a0 + RS + q * RPG * RS), acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], *(const bf16x8*)(a0 + 2 * RS + q * RPG * RS), acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3], *(const bf16x8*)(a3 + q * RPG * RS), acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4], *(const bf16x8*)(a4 + q * a4step), acc[q], 0, 0, 0);
            unsigned p00 = relu_pk(pack2(acc[0][0], acc[0][1])), p01 = relu_pk(pack2(acc[0][2], acc[0][3]));
            unsigned p10 = relu_pk(pack2(acc[1][0], acc[1][1])), p11 = relu_pk(pack2(acc[1][2], acc[1][3]));
            if (!interior) {
                const bool cin = (unsigned)(gx0 + col) < (unsigned)W;
                const bool in0 = cin && (unsigned)(gy0 + nr) < (unsigned)H, in1 = cin && (unsigned)(gy0 + RPG + nr) < (unsigned)H;
                p00 = in0 ? p00 : 0u; p01 = in0 ? p01 : 0u; p10 = in1 ? p10 : 0u; p11 = in1 ? p11 : 0u;
            }
            const u32x4 o = c16_pair_chunk(p00, p01, p10, p11);
            // lane (n, g): row (g & 1) * RPG + n of the strip column, chunk g >> 1
            if (n < RPG) *(u32x4*)(s_out + ((nr + (g & 1) * RPG + OOFF) * ORW + col + OOFF) * C16_PB + (((g >> 1) ^ (((col + OOFF) >> 2) & 1)) << 4)) = o;
        };
        constexpr int NPMAX = PA > PY1 ? (PA > PY2 ? PA : PY2) : (PY1 > PY2 ? PY1 : PY2);
#pragma unroll
        for (int k = 0; k < NPMAX; ++k)
            if (k < np) pair(p0 + k);
    }
#endif
}

// The last conv of the chain (region TH x TW = 24 x 32: two full column groups, three rows per wave) and what rides in its epilogue.
//   MODE 4      y = relu(conv + block input)                                  (plain chain)
//   MODE 1..3   y as above (stored when a.y is set); ysum = relu(bf16(y) + sum of MODE fuse-layer terms)
//   MODE 7      nothing stored but the heatmap: head(relu(relu(conv + block input) + 3 terms)) in fp32, argmax partial per tile
// The terms (slices of the 1x1-conv'd lower branches at 1/2, 1/4, 1/8 resolution, [pixel][16 channels] in LDS) enter through an MFMA
// whose A operand is the identity and whose C operand is the fp32 value they are added to: lane groups 0-1 feed the two 8-channel
// chunks of term 1, groups 2-3 those of term 2; a second MFMA (groups 0-1 only) adds term 3.
template <int RWI, int TH, int TW, int RWR, int ROFF, int MODE>
__device__ __forceinline__ void c16_conv_out(const char* s_in, const char* s_res, const char* s_terms, const BBFrag16& fr, bf16x8 idm, f32x4 hw4,
                                             const BBArgs& a, int gy0, int gx0, int b, int wave, int lane, BBBest* best) {
    static_assert(TW == 32 && TH % 8 == 0, "two full column groups");
    constexpr int RS = RWI * C16_PB, RSR = RWR * C16_PB;
    constexpr int RB = TH / 8;
    constexpr int NS = MODE == 7 ? 3 : (MODE == 4 ? 0 : MODE);
    constexpr int T2OFF = (TH >> 1) * (TW >> 1) * C16_PB, T3OFF = T2OFF + (TH >> 2) * (TW >> 2) * C16_PB;      // byte offsets of the slices of terms 2 and 3
    const int H = a.H, W = a.W;
    const int n = lane & 15, g = lane >> 4, h = g >> 1, c8 = g & 1;
    bf16x8 af[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) af[s] = fr.af[s];
    if (g >= 2) af[4] = idm;
    const f32x4 bias = fr.bias;
    // A operands of the term MFMAs: the identity on every lane group (two terms), on groups 0-1 only (one term)
    bf16x8 id_lo = idm;
    if (g >= 2) id_lo = bf16x8{};
    const bool interior = gy0 >= 0 && gy0 + TH <= H && gx0 >= 0 && gx0 + TW <= W;
    const int yb = wave * RB;
    const char* rowb = s_in + (yb * RWI) * C16_PB;
    const char* pA = rowb + c16_col(n + h, c8);
    const char* pC = rowb + h * RS + c16_col(n + 2, c8);
    const char* pD = rowb + 2 * RS + c16_col(n + 2, c8);
    int dstep = RS;
    if (h) { pD = s_res + ((yb + ROFF) * RWR) * C16_PB + c16_col(ROFF + n, c8); dstep = RSR; }
    const bool cin0 = (unsigned)(gx0 + n) < (unsigned)W, cin1 = (unsigned)(gx0 + 16 + n) < (unsigned)W;
    // lane masks, set up once (on interior tiles everything is live): column group 0 / 1 of this lane's pixel, the pixel of the
    // lane's 16-byte store (group g & 1), the lanes that hold heatmap values
    const bool cl0 = interior || cin0, cl1 = interior || cin1;
    const bool clw = (g & 1) ? cl1 : cl0;
    const bool hl0 = g == 0 && cl0, hl1 = g == 0 && cl1;
    const unsigned st_w = (unsigned)(((g & 1) * 16 + n) * C16_PB + (g >> 1) * 16);      // lane's 16 bytes of a 32-pixel run of 16-channel records
    // term fragments: per-lane base of column group 0 (group 1: + 16 >> shift pixels), the row part per row below
    const bool t2lane = NS >= 2 && g >= 2;               // this lane feeds term 2 (shift 2) into the first term MFMA; otherwise term 1 (shift 1)
    // slice row of output row y: (y >> 1) * (TW >> 1) or (y >> 2) * (TW >> 2) pixel records = (y & ~1) << 8 or (y & ~3) << 6 bytes for TW = 32
    const int tmask = t2lane ? ~3 : ~1, tshl = t2lane ? 6 : 8;
    const char* tbx[2];
    tbx[0] = s_terms + (t2lane ? T2OFF + (n >> 2) * C16_PB : (n >> 1) * C16_PB) + c8 * 16;
    tbx[1] = tbx[0] + (t2lane ? 4 * C16_PB : 8 * C16_PB);
    const char* tb3 = s_terms + T3OFF + (n >> 3) * C16_PB + c8 * 16;
#ifdef TTUP_C16_NARROW
    const unsigned st_16 = (unsigned)((n * 16 + g * 4) * 2);             // lane's 8 bytes inside a 16-pixel run of 16-channel records
#endif
    const float head_one = (n == 0) ? 1.f : 0.f;          // A operand of the head's cross-lane sum (row 0 of a 16x4 matrix of ones)
    float hvs[2 * RB];
    bf16x8 fa[2][RB + 2];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) { fa[xt][0] = *(const bf16x8*)(pA + xt * 512); fa[xt][1] = *(const bf16x8*)(pA + RS + xt * 512); }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int y = yb + r, gy = gy0 + y;
        bf16x8 f3[2], f4[2], ft[2], ft3[2];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            fa[xt][r + 2] = *(const bf16x8*)(pA + (r + 2) * RS + xt * 512);
            f3[xt] = *(const bf16x8*)(pC + r * RS + xt * 512);
            f4[xt] = *(const bf16x8*)(pD + r * dstep + xt * 512);
            if (NS >= 1) ft[xt] = *(const bf16x8*)(tbx[xt] + ((y & tmask) << tshl));
            if (NS >= 3) ft3[xt] = *(const bf16x8*)(tb3 + (y >> 3) * ((TW >> 3) * C16_PB) + xt * 2 * C16_PB);
        }
        f32x4 acc[2];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], fa[xt][r], bias, 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], fa[xt][r + 1], acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], fa[xt][r + 2], acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3], f3[xt], acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4], f4[xt], acc[xt], 0, 0, 0);
        const bool row_in = interior || (unsigned)gy < (unsigned)H;      // (wave-uniform)
        const size_t rowpix = (size_t)(b * H + gy) * W + gx0;          // (wave-uniform) first pixel of the tile's row in the image
        if (MODE != 7) {
            unsigned p[2][2], q[2][2];
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                p[xt][0] = relu_pk(pack2(acc[xt][0], acc[xt][1])); p[xt][1] = relu_pk(pack2(acc[xt][2], acc[xt][3]));
                if (NS >= 1) {
                    // the fuse sum starts from the ROUNDED branch output, what the element-wise pass reads back from memory
                    f32x4 ys = {bf16_to_f32((bf16_t)(p[xt][0] & 0xffff)), bf16_to_f32((bf16_t)(p[xt][0] >> 16)),
                                bf16_to_f32((bf16_t)(p[xt][1] & 0xffff)), bf16_to_f32((bf16_t)(p[xt][1] >> 16))};
                    ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(NS >= 2 ? idm : id_lo, ft[xt], ys, 0, 0, 0);
                    if (NS >= 3) ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(id_lo, ft3[xt], ys, 0, 0, 0);
                    q[xt][0] = relu_pk(pack2(ys[0], ys[1])); q[xt][1] = relu_pk(pack2(ys[2], ys[3]));
                }
            }
#ifdef TTUP_ABL_C4_NOSTORE
            asm volatile("" :: "v"(p[0][0]), "v"(p[0][1]), "v"(p[1][0]), "v"(p[1][1]));
            if (NS >= 1) asm volatile("" :: "v"(q[0][0]), "v"(q[0][1]), "v"(q[1][0]), "v"(q[1][1]));
#elif defined(TTUP_C16_NARROW)
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const bool live = row_in && (xt ? cl1 : cl0);
                if (a.y && live) *(u32x2*)((char*)(a.y + rowpix * 16) + (opaque_u32(st_16) + (unsigned)(xt * 512))) = u32x2{p[xt][0], p[xt][1]};
                if (NS >= 1 && live) *(u32x2*)((char*)(a.ysum + rowpix * 16) + (opaque_u32(st_16) + (unsigned)(xt * 512))) = u32x2{q[xt][0], q[xt][1]};
            }
#else
            // the two column groups' 8-byte pieces become ONE 16-byte store per lane (chunk g >> 1 of pixel n of group g & 1): half the
            // store instructions for the same bytes -- the store path is paid per instruction
            // (the swaps run with every lane active, BEFORE the divergent stores: a live lane's partner may be dead)
            const bool live = row_in && clw;
            const u32x4 yo = c16_pair_chunk(p[0][0], p[0][1], p[1][0], p[1][1]);
            u32x4 so = yo;
            if (NS >= 1) so = c16_pair_chunk(q[0][0], q[0][1], q[1][0], q[1][1]);
            if (a.y && live) *(u32x4*)((char*)(a.y + rowpix * 16) + opaque_u32(st_w)) = yo;
            if (NS >= 1 && live) *(u32x4*)((char*)(a.ysum + rowpix * 16) + opaque_u32(st_w)) = so;
#endif
        } else {
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                // neither the branch tensor nor the sum is stored: no rounding in front of the head
                f32x4 ys = {relu_f32(acc[xt][0]), relu_f32(acc[xt][1]), relu_f32(acc[xt][2]), relu_f32(acc[xt][3])};
                ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idm, ft[xt], ys, 0, 0, 0);
                ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(id_lo, ft3[xt], ys, 0, 0, 0);
                float part = relu_f32(ys[0]) * hw4[0];
                part = fmaf(relu_f32(ys[1]), hw4[1], part);
                part = fmaf(relu_f32(ys[2]), hw4[2], part);
                part = fmaf(relu_f32(ys[3]), hw4[3], part);
                // sum over the pixel's 4 lane groups on the matrix pipe (exact fp32): D[0][n] = sum_g 1 * part(n, g)
                const f32x4 hd = __builtin_amdgcn_mfma_f32_16x16x4f32(head_one, part, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const float hv = hd[0] + a.hbias;
#ifdef TTUP_ABL_C4_NOSTORE
                asm volatile("" :: "v"(hv));
#else
                if (row_in && (xt ? hl1 : hl0)) *(float*)((char*)(a.heat + rowpix) + (opaque_u32((unsigned)(n * 4)) + (unsigned)(xt * 64))) = hv;
#endif
                hvs[2 * r + xt] = hv;
            }
        }
    }
    if (MODE == 7) {
        // argmax partial of the lane's 2 * RB heatmap values (lanes g == 0 hold them).  The groups come in index order, so "greater
        // value, then lower index" is ONE strict compare per group -- unless a live value is NaN or -inf (torch.argmax: a NaN wins, ties
        // go to the lower index even at -inf): then the careful comparison runs over the values again (never on real heatmaps).
        float bv = -INFINITY; int bg = -1;
        bool odd = false;
#pragma unroll
        for (int k = 0; k < 2 * RB; ++k) {
            const bool live = interior || ((unsigned)(gy0 + yb + (k >> 1)) < (unsigned)H && ((k & 1) ? cin1 : cin0));
            const bool gt = hvs[k] > bv;
            if (live && gt) { bv = hvs[k]; bg = k; }
            odd = odd || (live && !(hvs[k] > -INFINITY));
        }
        best->v = bv;
        best->i = 0x7fffffffffffffffLL;
        if (bg >= 0) best->i = (long long)(gy0 + yb + (bg >> 1)) * W + gx0 + (bg & 1) * 16 + n;
        if (__builtin_amdgcn_ballot_w64(odd && g == 0) != 0ull) {
            best->v = -INFINITY; best->i = 0x7fffffffffffffffLL;
#pragma unroll
            for (int k = 0; k < 2 * RB; ++k) {
                const int gy = gy0 + yb + (k >> 1);
                const bool live = interior || ((unsigned)gy < (unsigned)H && ((k & 1) ? cin1 : cin0));
                const long long e = (long long)gy * W + gx0 + (k & 1) * 16 + n;
                if (live && bb_better(hvs[k], e, best->v, best->i)) { best->v = hvs[k]; best->i = e; }
            }
        }
    }
}

// One tile per workgroup, two workgroups per CU (LDS 2 x 79.5 KB, at most 128 VGPRs); the weights come straight from L2 into registers.
template <int TH, int TW, int MODE>
__global__ __launch_bounds__(512, 4) void c16_chain_kernel(BBArgs a) {
    constexpr int C = 16, L = 4;
    constexpr int R0H = TH + 2 * L, R0W = TW + 2 * L;
    // row strides (pixels) of the two LDS buffers: ODD, so that 16 consecutive rows of one column fall on 16 different 16-byte slots (strip pairs)
    constexpr int SA = (R0W & 1) ? R0W : R0W + 1;
    constexpr int SB = ((R0W - 2) & 1) ? R0W - 2 : R0W - 1;
    constexpr int S3 = R0W - 6;                          // conv3's output rows are packed: the tail of bufB behind them holds the fuse-term slices
    constexpr int SZ_A = R0H * SA * C16_PB, SZ_B = (R0H - 2) * SB * C16_PB;      // bytes
    constexpr int NS = MODE == 7 ? 3 : (MODE == 4 ? 0 : MODE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bufA = smem;
    char* bufB = smem + SZ_A;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 3-D grid (tile column, tile row, image); XCD = linear workgroup id % 8: every XCD takes a strip of adjacent tile columns (see bb_chain2_kernel)
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
#if defined(TTUP_TIMING) && defined(TTUP_TIMING_C16W)
    unsigned long long wst[16] = {};
#endif
    C16_WSTAMP(0);
    BBFrag16 fr;
    bf16x8 idm;
    f32x4 hw4 = {0.f, 0.f, 0.f, 0.f};
    {
        // the tile with its 4-pixel halo: a thread keeps one 16-byte column unit and walks rows (row lane rl, then every RL-th row); all
        // of its loads are issued before the first LDS store -- ONE memory round trip for the tile
        constexpr int CU = R0W * (C / 8);                 // 16-byte units per tile row
        constexpr int RL = 512 / CU;                      // row lanes
        constexpr int IN_PT = (R0H + RL - 1) / RL;
        static_assert(RL >= 1, "tile row wider than the workgroup");
        const int cu = tid % CU, rl = tid / CU;
        const int col = cu / (C / 8), c8 = cu % (C / 8);
        u32x4 v[IN_PT];
        const bool halo_in = oy0 >= L && oy0 - L + R0H <= a.H && ox0 >= L && ox0 - L + R0W <= a.W;      // workgroup-uniform: 89 % of the tiles at 704 x 1280
        if (halo_in) {
            // scalar row bases + one per-lane byte offset, no bounds tests.  Lanes without a unit (rl == RL, or past the last row in the
            // last pass) fetch a unit that exists and store nothing.
            const char* tb = (const char*)(a.x + ((long long)(b * a.H + oy0 - L) * a.W + (ox0 - L)) * C);
            const int rlc = rl < RL ? rl : RL - 1;
            const unsigned voff = (unsigned)((rlc * a.W + col) * C + c8 * 8) * 2u;
            // the last pass has rows for some row lanes only: the others fetch the tile's last row again (offset from the first pass's base)
            const int row_last = rlc + (IN_PT - 1) * RL < R0H ? rlc + (IN_PT - 1) * RL : R0H - 1;
            const unsigned voff_last = (unsigned)((row_last * a.W + col) * C + c8 * 8) * 2u;
#pragma unroll
            for (int k = 0; k < IN_PT; ++k)
                v[k] = k == IN_PT - 1 ? *(const u32x4*)(tb + opaque_u32(voff_last)) : *(const u32x4*)(tb + (size_t)k * RL * a.W * C * 2 + opaque_u32(voff));
        } else {
            const int gx = ox0 - L + col, gyb = oy0 - L + rl;
            const bool col_ok = rl < RL && gx >= 0 && gx < a.W;
            const bf16_t* src = a.x + ((long long)(b * a.H + gyb) * a.W + gx) * C + c8 * 8;      // may point outside for halo rows / columns: only dereferenced when valid
            const long long row_step = (long long)RL * a.W * C;
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) {
                const int gy = gyb + k * RL;
                // branch-free: an invalid unit reads the tensor's first bytes and is zeroed afterwards
                const bool ok = col_ok && rl + k * RL < R0H && gy >= 0 && gy < a.H;
                const u32x4 t = *(const u32x4*)(ok ? src + k * row_step : a.x);
                v[k] = u32x4{ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u};
            }
        }
        // requested right behind the tile (L2 hits: they are there when the tile is, instead of costing a second round trip behind the
        // first barrier): the first conv's weight fragments, the identity fragment (a table), the head weights
        bb_load_frag16(fr, a.w[0], a.bias[0], lane);
        idm = *(const bf16x8*)(c16_idm_tab.v + lane * 8);
        if (MODE == 7) hw4 = *(const f32x4*)(a.hw + (lane >> 4) * 4);
        char* dst = bufA + (rl * SA) * C16_PB + c16_col(col, c8);
#pragma unroll
        for (int k = 0; k < IN_PT; ++k)
            if (rl < RL && rl + k * RL < R0H) *(u32x4*)(dst + k * RL * SA * C16_PB) = v[k];
    }
    C16_WSTAMP(1);
    __syncthreads();
    TTUP_STAMP(1);
    C16_WSTAMP(2);
#if TTUP_C16_PREFETCH > 0
    unsigned pf_sink = 0;          // the register the prefetch lands in: kept alive (below) until the load has certainly arrived -- the compiler does not know about it
    // L2 prefetch for the workgroup that takes this one's place: workgroups are dealt to the XCDs round-robin by linear id and 512 are
    // resident (two per CU), so workgroup id + TTUP_C16_PREFETCH (a multiple of 8: same XCD, same L2) starts about one tile time from
    // now.  One lane per 128-byte line of ITS input region reads one dword -- 320 lane-loads for 40 KB -- into a register nobody
    // reads (an asm the compiler does not count: its later waits can only get stricter, and s_endpgm waits for everything).
    {
        const int lid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x + TTUP_C16_PREFETCH;
        const int gxy = gridDim.x * gridDim.y;
        const int pz = lid / gxy, prem = lid - pz * gxy, py = prem / (int)gridDim.x, pxr = prem - py * (int)gridDim.x;
#ifdef TTUP_NO_XCD_MAP
        const int pbx = pxr;
#else
        const int pbx = (gridDim.x & 7) == 0 ? (pxr & 7) * (gridDim.x >> 3) + (pxr >> 3) : pxr;
#endif
        constexpr int LPR = (R0W * C16_PB + 127) / 128;          // 128-byte lines per region row (the region starts 128-byte aligned: ox0 - 4 pixels of 32 bytes)
        const int prow = tid / LPR, pline = tid - prow * LPR;
        const int gyp = py * TH - L + prow, gxp = pbx * TW - L + pline * 4;          // 4 pixels per line
        if (pz < (int)gridDim.z && prow < R0H && gyp >= 0 && gyp < a.H && gxp >= 0 && gxp < a.W) {
            const bf16_t* pp = a.x + ((long long)(pz * a.H + gyp) * a.W + gxp) * C;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(pp) : "memory");
        }
    }
#endif
    c16_conv_lds<SA, 0, R0H - 2, R0W - 2, false, 1, 0, SB, 0, C16_SPLIT1>(bufA, bufB, nullptr, fr, idm, oy0 - 3, ox0 - 3, a.H, a.W, wave, lane);
    TTUP_STAMP(2);
    C16_WSTAMP(3);
    // next conv's fragments: requested BEFORE the barrier, in flight across it.  (Requested a whole conv earlier into a second register
    // set -- 110-116 instead of 86-100 VGPRs -- the kernel is no faster: 0.7311 against 0.7310 ms for its three launches.)
    bb_load_frag16(fr, a.w[1], a.bias[1], lane);
    __syncthreads();
    TTUP_STAMP(3);
    C16_WSTAMP(4);
    c16_conv_lds<SB, 0, R0H - 4, R0W - 4, true, SA, 2, SA, 2, C16_SPLIT2>(bufB, bufA, bufA, fr, idm, oy0 - 2, ox0 - 2, a.H, a.W, wave, lane);
    C16_WSTAMP(5);
#if TTUP_C16_PREFETCH > 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (everything older has long arrived: conv1 and conv2 lie between)
    asm volatile("" :: "v"(pf_sink));
#endif
    bb_load_frag16(fr, a.w[2], a.bias[2], lane);
    __syncthreads();
    TTUP_STAMP(4);
    C16_WSTAMP(6);
    // The fuse-layer terms of the last epilogue (1x1-conv'd lower branches at 1/2, 1/4, 1/8 resolution): the tile's slices (12x16 + 6x8 +
    // 3x4 pixel records = 8 KB at most) are requested now, travel while conv3 runs, and are parked in the tail of bufB that conv3's
    // packed 26x34 output leaves free.  One branch-free load per thread (a unit outside the image reads the term's first bytes: it is
    // only ever added to outputs that are not stored).
    constexpr int T_OFF = (TH + 2) * S3 * C16_PB;          // bytes of bufB taken by conv3's output
    static_assert(TH % 8 == 0 && TW % 8 == 0, "term slices are aligned to the tile for 8-aligned tiles");
    static_assert(SZ_B - T_OFF >= ((TH >> 1) * (TW >> 1) + (TH >> 2) * (TW >> 2) + (TH >> 3) * (TW >> 3)) * C16_PB, "bufB tail holds the term slices");
    char* s_terms = bufB + T_OFF;
    u32x4 treg = u32x4{0u, 0u, 0u, 0u};
    constexpr int B1 = (TH >> 1) * (TW >> 1) * 2, B2 = B1 + (TH >> 2) * (TW >> 2) * 2, B3 = B2 + (TH >> 3) * (TW >> 3) * 2;      // 16-byte units
    constexpr int BN = NS == 0 ? 0 : NS == 1 ? B1 : NS == 2 ? B2 : B3;
    if (NS > 0) {
        const int k = tid < B1 ? 0 : tid < B2 ? 1 : 2, sh = k + 1;
        const int u = tid - (k == 0 ? 0 : k == 1 ? B1 : B2);
        const int px = u >> 1;
        const int pr = k == 0 ? px / (TW >> 1) : k == 1 ? px / (TW >> 2) : px / (TW >> 3);
        const int pc = px - pr * (TW >> sh);
        const int ty = (oy0 >> sh) + pr, tx = (ox0 >> sh) + pc;
        const int hs = a.H >> sh, ws = a.W >> sh;
        const bf16_t* tp = k == 0 ? a.st[0] : k == 1 ? a.st[1] : a.st[2];
        const bool ok = tid < BN && ty < hs && tx < ws;
        treg = *(const u32x4*)(ok ? tp + ((long long)(b * hs + ty) * ws + tx) * 16 + (u & 1) * 8 : a.st[0]);
    }
    c16_conv_lds<SA, 2, R0H - 6, R0W - 6, false, 1, 0, S3, 0, C16_SPLIT3>(bufA, bufB, nullptr, fr, idm, oy0 - 1, ox0 - 1, a.H, a.W, wave, lane);
    C16_WSTAMP(7);
    if (NS > 0 && tid < BN) ((u32x4*)s_terms)[tid] = treg;
    bb_load_frag16(fr, a.w[3], a.bias[3], lane);
    __syncthreads();
    TTUP_STAMP(5);
    C16_WSTAMP(8);
    BBBest best; best.v = -INFINITY; best.i = 0x7fffffffffffffffLL;
    c16_conv_out<S3, TH, TW, SA, 4, MODE>(bufB, bufA, s_terms, fr, idm, hw4, a, oy0, ox0, b, wave, lane, &best);
#ifdef TTUP_TIMING_SPLIT
    TTUP_STAMP(6);
#endif
    C16_WSTAMP(9);
    if (MODE == 7) {
        // argmax partial of this tile.  (value, index) pairs become one 64-bit key -- order-preserving bits of the value (NaN on top,
        // -0 = +0 as torch.argmax has it) above the complemented index -- so that "greater value, then lower index" is an unsigned
        // max: 4 DPP row shifts over the 16 lanes that hold heatmap values, one LDS slot per wave (behind the tile buffers: no
        // barrier before writing it), one barrier, 3 more shifts in wave 0.
        unsigned long long key = 0ull;           // below every real key
        if (lane < 16 && best.i != 0x7fffffffffffffffLL) key = bb_key(best.v, (int)best.i);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) { const unsigned long long o = bb_dpp_shl(key, off); key = o > key ? o : key; }
        unsigned long long* slots = (unsigned long long*)(smem + (size_t)(SZ_A + SZ_B));
        if (lane == 0) slots[wave] = key;
        __syncthreads();
        if (wave == 0) {
            key = lane < 8 ? slots[lane] : 0ull;
#pragma unroll
            for (int off = 4; off >= 1; off >>= 1) { const unsigned long long o = bb_dpp_shl(key, off); key = o > key ? o : key; }
            if (lane == 0) {
                a.pv[(size_t)b * a.tiles_per_img + tt] = bb_key_value(key);
                a.pi[(size_t)b * a.tiles_per_img + tt] = (long long)(~(unsigned)key);
            }
        }
    }
#ifndef TTUP_TIMING_SPLIT
    TTUP_STAMP(6);
#endif
#if defined(TTUP_TIMING) && defined(TTUP_TIMING_C16W)
    C16_WSTAMP(10);
    __builtin_amdgcn_s_waitcnt(0);           // (vmcnt 0: the tile's stores have left)
    C16_WSTAMP(11);
    if (lane == 0 && TTUP_BID < 512) {
#pragma unroll
        for (int k = 0; k < 12; ++k) ttup_tbuf[(TTUP_BID * 8 + wave) * 16 + k] = wst[k];
    }
#elif defined(TTUP_TIMING)
    if (tid == 0 && TTUP_BID < 8192) ttup_tbuf[TTUP_BID * 8 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;      // 100 MHz ticks for the same span
#endif
}

template <int TH, int TW, int MODE>
static int launch_c16_t(const BBArgs& a, int batch, int h, int w, hipStream_t st) {
    constexpr int SA = ((TW + 8) & 1) ? TW + 8 : TW + 9, SB = ((TW + 6) & 1) ? TW + 6 : TW + 7;       // odd row strides, as in the kernel
    constexpr size_t SMEM = (size_t)((TH + 8) * SA + (TH + 6) * SB) * C16_PB + 64;       // + one argmax slot per wave
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
    if (int rc = ensure_max_lds((const void*)c16_chain_kernel<TH, TW, MODE>, SMEM)) return rc;
    BBArgs k = a;
    k.H = h; k.W = w; k.tiles_x = cdiv(w, TW); k.tiles_per_img = k.tiles_x * cdiv(h, TH); k.total_tiles = k.tiles_per_img * batch;
    if (k.total_tiles == 0) return TTUP_OK;
    kernel_note("c16_chain_kernel<%d, %d, %d>", TH, TW, MODE);
    hipLaunchKernelGGL((c16_chain_kernel<TH, TW, MODE>), dim3(k.tiles_x, cdiv(h, TH), batch), dim3(512), SMEM, st, k);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}
