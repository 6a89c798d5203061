// Convolution + pointwise kernels of the WASB/HRNet CNN (reference balldetection/models/wasb.py).
// Activations are NHWC; two arithmetic modes:
//   bf16 : bf16 storage, v_mfma_f32_16x16x32_bf16, fp32 accumulate   (production path)
//   f32  : fp32 storage, direct fp32 FMA                               (parity / debug path)
#pragma once
#include "common.h"
#include <vector>

namespace ttup {

// One convolution with eval-mode BatchNorm folded in (scale into the weights, shift into the bias).
struct FoldedConv {
    int cout = 0, cin = 0, k = 1, stride = 1;
    std::vector<float> w;      // [cout][cin][k][k]
    std::vector<float> bias;   // [cout]
};

// Device-side packed weights for one conv op (possibly two folded convs concatenated along K).
struct PackedConv {
    int cout = 0, cin_total = 0, c0 = 0, k = 1, stride = 1, ck = 32;
    void*  w_dev = nullptr;     // bf16 MFMA fragments, or f32 [tap][cin][cout]
    void*  w3_dev = nullptr;    // f32 handles: the weights split into three bf16 planes, per MFMA fragment (csrc/conv_x3.hip); null when the shape is not covered
    int    mt3 = 0;             // couts per block of that packing / 16
    float* bias_dev = nullptr;  // [cout]
    size_t w_bytes = 0;
};

// Part of a tensor that an op has to produce: rows [y0, y1), columns [x0, x1) of sample b, applied only where flag[b] != 0 (device
// memory, one int per sample of the pass).  The fp32 crop net of the certified argmax is pruned to the cone of its 24-pixel core (csrc/wasb_net.hip compute_roi): a
// kernel may compute any superset of the region -- what lies outside is never read by an op that matters.  flag == nullptr: no pruning.
// Two classes of pruned samples (round 6): flag[b] == 1 -> [y0, y1) x [x0, x1) (crops with the 24-pixel core), flag[b] == 2 -> the `s`
// rectangle (crops whose candidates fit a 16-pixel core in the crop's corner-aligned 160 x 160 part: one 16-pixel tile column / row
// less in every full-resolution layer).
struct Roi {
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    int sy0 = 0, sy1 = 0, sx0 = 0, sx1 = 0;          // class 2 (sy1 == 0: same as class 1)
    const int* flag = nullptr;
    // true when pixel (y, x) of a sample with flag value f (!= 0) lies outside what the op has to produce
    __host__ __device__ bool outside(int f, int y, int x) const {
        if (f == 2 && sy1 > 0) return y < sy0 || y >= sy1 || x < sx0 || x >= sx1;
        return y < y0 || y >= y1 || x < x0 || x >= x1;
    }
};

struct ConvLaunch {
    const void* src0 = nullptr;   // NHWC, c0 channels
    const void* src1 = nullptr;   // NHWC, cin_total-c0 channels (two-source 1x1 only) or null
    const void* residual = nullptr;  // NHWC cout channels at output resolution, or null
    void* dst = nullptr;          // NHWC cout channels
    int batch = 0, h = 0, w = 0;  // input spatial size
    int relu = 0;
    const PackedConv* follow = nullptr;   // bf16 only: fused 1x1 follower (64 -> 32, ReLU) applied to dst, written to dst2
    void* dst2 = nullptr;
    const void* res2 = nullptr;           // bf16 stride-2 convs: further fuse-layer terms added before the ReLU (same resolution /
    const void* res3 = nullptr; int sh3 = 0;   // 1/2^sh3 resolution, nearest-neighbour upsampled), wasb.py:236-243
    const int* n_active = nullptr;        // f32 only: device-side batch (<= batch) decided by an earlier kernel (csrc/certify.hip)
    Roi roi;                              // f32 split-bf16 kernels only: output region to produce (cone pruning of the crop net)
    // bf16 64 -> 64 3x3 only: linear 1x1 followers (fuse-layer convs 64 -> 16 / 64 -> 32, no ReLU) applied to dst
    const PackedConv* lin16 = nullptr; void* lin16_dst = nullptr;
    const PackedConv* lin32 = nullptr; void* lin32_dst = nullptr;
    // bf16 3x3 stride-2 16 -> 32 only: a second stride-2 conv 16 -> 16 on the same input, run in the same pass (one read of src0)
    const PackedConv* pair = nullptr; void* pair_dst = nullptr; int pair_relu = 0;
};

// host-side packing (called from ttup_wasb_create)
int pack_conv(const FoldedConv& a, const FoldedConv* b /*second source or null*/, int cin_pad, int dtype, PackedConv* out);
void free_conv(PackedConv* p);

int launch_conv(const PackedConv& p, const ConvLaunch& l, int dtype, hipStream_t stream);
// fp32 matrix-pipe conv (csrc/conv_f32.hip): exact fp32 products on v_mfma_f32_16x16x4_f32 (cross-check path, TTUP_F32_EXACT=1)
bool conv_f32_mfma_supported(const PackedConv& p);
int launch_conv_f32_mfma(const PackedConv& p, const ConvLaunch& l, hipStream_t stream);
// fp32 conv on the bf16 matrix pipe with operands split into three bf16 parts (csrc/conv_x3.hip): the fp32 path's kernel
bool conv_x3_supported(const PackedConv& p);
int launch_conv_x3(const PackedConv& p, const ConvLaunch& l, hipStream_t stream);
int pack_conv_x3(const std::vector<float>& w_tap_cin_cout, int cout, int cin_total, int c0, int k, int stride, PackedConv* out);

// fused stem: conv1 + conv2 + Bottleneck conv1, bf16 only, persistent with all weights resident in LDS (csrc/conv.hip)
// frames_per_sample = 0: x0 is the (B,H,W,16) input tensor; 1 / 3: x0 is the per-frame pre-processed clip (B+nf-1,H,W,4) and p1 is
// packed in the slot order f*4 + c (see stem_kernel)
int launch_stem(const PackedConv& p1, const PackedConv& p2, const PackedConv& p3, const void* x0, void* t2, void* a1,
                int batch, int h, int w, hipStream_t st, int frames_per_sample = 0);
#define TTUP_LAYOUT_NHWC4_FRAME 2      // internal: launch_preprocess output = one bf16 (c0,c1,c2,0) record per frame pixel

// fused Bottleneck tail (conv3 + downsample + add + relu) + both transition1 convs, bf16 only (csrc/conv.hip)
int launch_bneck_trans(const PackedConv& p1, const PackedConv& p5, const PackedConv& p6, const void* a2, const void* t2,
                       void* b0, void* b1, int batch, int h, int w, hipStream_t st);

// fused chain of 1 or 2 BasicBlocks (2 or 4 3x3 convs, C = 16 or 32) of one branch, bf16 only (csrc/conv.hip).
// BBSum (16-channel two-block chain only): the fuse-layer sum that consumes the branch, computed in the epilogue of the last
// conv -- ysum = relu(y + sum_k up(terms[k], 2^shifts[k])); with `heat` set the sum is not stored: the 1x1 head and the
// per-tile argmax partial (pv / pi [map * bb_chain_tiles_per_img + tile]) are computed from it in registers, and y may be null.
struct BBSum {
    const void* terms[3] = {nullptr, nullptr, nullptr}; int shifts[3] = {0, 0, 0}; int n_terms = 0;
    void* ysum = nullptr;
    float* heat = nullptr; const float* head_w = nullptr; float head_bias = 0.f; float* pv = nullptr; long long* pi = nullptr;
};
int bb_chain_tiles_per_img(int h, int w);
int launch_bb_chain(const PackedConv* const* convs, int n_convs, const void* x, void* y, int batch, int h, int w,
                    const PackedConv* follow, void* y_follow, hipStream_t st, const BBSum* sum = nullptr);

// y = relu(base + sum_k nearest_upsample(t_k, 2^shift_k)); all NHWC with c channels; base at (h,w).
int launch_upsum(const void* base, const void* const* terms, const int* shifts, int n_terms, void* dst,
                 int batch, int h, int w, int c, int dtype, hipStream_t stream, const int* n_active = nullptr, const Roi* roi = nullptr);

// float32 NCHW (B,cin,H,W) -> NHWC with cpad channels (zero filled)
int launch_nchw_to_nhwc(const float* src, void* dst, int batch, int cin, int cpad, int h, int w, int dtype, hipStream_t stream);
// NHWC (any dtype) -> float32 NCHW (debug taps)
int launch_nhwc_to_nchw(const void* src, float* dst, int batch, int c, int h, int w, int dtype, hipStream_t stream);

// head: 1x1 conv cin -> n_out output channels (+bias), float32 (B,n_out,H,W) out; w_dev [n_out][cin], bias_dev [n_out]
int launch_head(const void* src, const float* w_dev, const float* bias_dev, int n_out, float* heat, int batch, int h, int w, int cin,
                int dtype, hipStream_t stream, const int* n_active = nullptr, const Roi* roi = nullptr);

// uint8 frames -> normalised triples: see ttup_preprocess_triples.  out NCHW f32 or NHWC16 (dtype of the net)
int launch_preprocess(const uint8_t* frames, int n_frames, int src_h, int src_w, int dst_h, int dst_w,
                      void* out, int out_layout, int dtype, int first_triple, int n_triples, int frames_per_sample, hipStream_t stream);

int launch_preprocess_crops(const uint8_t* frames, int n_frames, int src_h, int src_w, int dst_h, int dst_w, float* out,
                            const int* crops_dev, int crop0, const int* n_active_dev, int max_crops, int crop_h, int crop_w,
                            int frames_per_sample, hipStream_t stream);

}  // namespace ttup
