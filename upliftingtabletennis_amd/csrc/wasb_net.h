// Internal definition of the CNN handle (ttup_wasb): a static op list over NHWC buffers, shared by csrc/wasb_net.hip (graph
// construction, forward) and csrc/certify.hip (certified argmax).  Not part of the C ABI.
#pragma once
#include "conv.h"
#include <map>
#include <string>
#include <vector>

namespace ttup {
size_t upsum_head_ws_bytes(int n_maps, int H, int W);
int launch_upsum_head(const void* base, const void* const* terms, const int* shifts, int n_terms, const float* w_dev, float bias,
                      float* heat, int n_maps, int H, int W, long long* argmax, float* win, void* ws, size_t ws_bytes, hipStream_t st);
int refine_argmax(const float* heat, int n_maps, int H, int W, long long* argmax, float* win, void* ws, size_t ws_bytes, hipStream_t st);
int launch_argmax_finish(const float* heat, int n_maps, int H, int W, int nblk, const float* pv, const long long* pi, long long* argmax, float* win, hipStream_t st);

struct Tensor { void* ptr = nullptr; int c = 0, h = 0, w = 0; int extra = 0; };     // (micro + extra, h, w, c)

struct Op {
    enum Kind { CONV, UPSUM, BNECK_TRANS, BB_CHAIN, UPSUM_HEAD, STEM } kind = CONV;
    int chain[4] = {-1, -1, -1, -1}, n_chain = 0;          // BB_CHAIN: packed conv indices
    int conv = -1;            // index into packed convs
    int conv2 = -1, conv3 = -1, dst2 = -1;     // BNECK_TRANS: transition convs and second output; CONV: fused 1x1 follower (conv2) -> dst2
    int src0 = -1, src1 = -1, residual = -1, dst = -1;
    int relu = 0;
    int terms[3] = {-1, -1, -1}, shifts[3] = {0, 0, 0}, n_terms = 0;   // UPSUM
    int res2 = -1, res3 = -1, sh3 = 0;          // CONV (bf16, stride 2): fuse-layer terms folded into the epilogue
    // BB_CHAIN (16 channels, 4 convs) with the consuming fuse-layer sum in its epilogue: terms/shifts/n_terms as for UPSUM,
    // dst2 = the summed output; head = 1: stage-4 output, never stored -- the 1x1 head + argmax partials are computed from it
    // (launched by run_head_op, which knows the output buffers); dst = -1 when the pre-fuse branch tensor has no consumer
    int head = 0;
    int conv1f = -1;                            // STEM: conv1 packed for the frames mode (channel slot f*4 + c)
    // CONV (64 -> 64 3x3, bf16): fuse-layer 1x1 convs on its output riding in its epilogue (packed conv index, output tensor)
    int lin16 = -1, lin16_dst = -1, lin32 = -1, lin32_dst = -1;
    // CONV (3x3 s2 16 -> 32, bf16): a second 3x3 s2 16 -> 16 conv on the same input in the same pass (packed conv, output, ReLU)
    int pair = -1, pair_dst = -1, pair_relu = 0;
};

// Certified argmax (csrc/certify.hip): state owned by a bf16 ball-detector handle
struct CertState {
    bool enabled = false;
    float eps = 0.f;                     // bound on |bf16 heatmap - fp32 heatmap| (absolute, calibrated by the caller)
    static constexpr float GUARD = 1.25f;   // a heatmap with an empty guard band stays certified when eps is widened by up to this factor
    int R = 72;                          // receptive-field radius of one heatmap pixel (measured: 71)
    int small = 0;                       // class-2 crops: candidates that fit the core positions R + 1 .. R + small get a crop pruned to that core's cone (0 = off)
    int K = 256;                         // candidates kept per heatmap (<= CERT_MAX_K, csrc/certify.hip): the flat top of a saturated blob fits
    int maxc = 8;                        // new crops a heatmap may add (ttup.h: max_crops_per_map, 0 = 8)
    int maxf = 8;                        // crops per frame, set by ttup_wasb_set_certify to min(16, maxc * channels): the channels of a frame share them
    int CH = 0, nchunks = 0, max_crops = 0, Hc = 0, Wc = 0;
    int budget = 0;                      // crops the next forward may use (<= max_crops): ceil(budget / CH) fp32 passes are enqueued
    bool exact_windows = false;          // every heatmap gets an fp32 crop (also single-candidate ones): all 3x3 windows are fp32 values
    int audit_mod = 0, audit_phase = 0;  // audit crops: single-candidate heatmaps of the frames with (frame + phase) % mod == 0 get an fp32 crop as well (0 = off)
    struct ::ttup_wasb* cropnet = nullptr;  // fp32 handle at crop size, batch CH
    // Per-call state, two slots used alternately: the fp32 passes of call k run on the handle's own stream (`stream`) while the
    // bf16 micro-batches of call k+1 -- issued on another caller stream -- already fill slot (k+1)&1
    struct Slot {
        int* cand_idx = nullptr; int* cand_cnt = nullptr; int* cand_crop = nullptr; float* cand_val = nullptr; float* cand_win = nullptr;
        float* cand_bf = nullptr;           // the bf16 path's value of every candidate (audit: |bf16 - fp32| at the candidates is free)
        int* guard_cnt = nullptr;           // pixels per heatmap in the guard band below the candidate band
        int* crop_rec = nullptr; int* n_crops = nullptr; int* n_active = nullptr; int* status = nullptr;
        int* roi_flag = nullptr;            // per crop: 1 = interior crop (the pruned op regions of the crop net apply to it)
        float* margin = nullptr;            // fp32 top-2 margin among the candidates of a resolved heatmap (+inf: one candidate / not resolved)
        hipEvent_t done = nullptr;          // fp32 passes of the call that last used the slot have finished
        // the caller's copies of this slot's status / crop count (ttup_wasb_certify_status / _flags / _info, on whatever stream the
        // caller issued them) have finished: the next call that takes the slot waits for them before it zeroes the slot (a call
        // issued on ANOTHER stream -- e.g. after an odd number of extra calls changed the stream / slot pairing -- would otherwise
        // reset the status under a pending copy; round-3 advisor, medium)
        hipEvent_t read_status = nullptr, read_info = nullptr, read_margin = nullptr;          // one event per KIND of copy: a record overwrites the event's previous record
    } slot[2];
    int cur = 0;                            // slot of the call being issued / last issued
    unsigned long long* stats = nullptr;
    float* crop_heat = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t lanes_done = nullptr;
};
}  // namespace ttup

struct ttup_wasb {
    int H = 0, W = 0, max_batch = 0, dtype = 0, micro = 1, in_ch = 9;
    std::vector<ttup::PackedConv> convs;
    std::vector<ttup::Tensor> tensors;
    std::vector<ttup::Op> ops;
    std::map<std::string, int> taps;
    int t_input = -1, t_out = -1;
    int t_frames = -1;                  // bf16 stem frames mode: (micro + nf - 1, H, W, 4) per-frame pre-processed records
    bool frames_mode = false;           // what the stem op reads in the current pass (set by forward_micro)
    float* head_w_dev = nullptr; float* head_b_dev = nullptr; float head_bias = 0.f;
    int n_out = 1;                      // heatmap channels returned: 1 (ball: channel 1 of 3, wasb.py:606) or all 13 (table, hrnet.py:586-589)
    float* heat_scratch = nullptr;      // (micro,H,W) when the caller does not want heatmaps
    void* refine_ws = nullptr; size_t refine_ws_bytes = 0;
    long long* argmax_scratch = nullptr; float* win_scratch = nullptr;
    int last_batch = 0;
    bool fused_head = false;      // last op computes the heatmap and the argmax partials itself (bf16 path)
    const int* n_active = nullptr;      // fp32 crop net of the certified argmax: device-side batch of the current pass
    // ... and its cone pruning: op_roi[k] = the region of op k's output that the crop's core depends on (compute_roi), out_roi = the
    // region of the heatmap itself; applied to the samples b with roi_flag[b] != 0 (interior crops; a crop that touches an image border has a core
    // that reaches that border and is computed in full).  Empty op_roi = no pruning (every other handle).
    std::vector<ttup::Roi> op_roi; ttup::Roi out_roi; const int* roi_flag = nullptr;
    std::vector<char> blob;             // the weight blob the handle was created from (the certified argmax builds its fp32 twin from it)
    ttup::CertState cert;
    // Lanes: independent micro-batches alternate between `lanes.size()` internal streams, each with its own activation
    // and scratch buffers, so one micro-batch's kernel tails and launch gaps are filled by the other's kernels.
    // `tensors[i].ptr` and the scratch pointers above always alias the lane in use (use_lane).
    struct Lane {
        std::vector<void*> ptr;
        float* heat_scratch = nullptr; void* refine_ws = nullptr; long long* argmax_scratch = nullptr; float* win_scratch = nullptr;
        hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    };
    std::vector<Lane> lanes;
    hipEvent_t fork = nullptr;
    // bf16 / fp32 micro-batches of consecutive forward calls share the lanes' activation and scratch buffers: a call waits for the
    // previous call's last micro-batch (whatever stream that call was issued on) before it touches them
    hipEvent_t pass_done = nullptr;
    bool pass_recorded = false;

    void use_lane(int l) {
        const Lane& L = lanes[l];
        for (size_t i = 0; i < tensors.size(); ++i) tensors[i].ptr = L.ptr[i];
        heat_scratch = L.heat_scratch; refine_ws = L.refine_ws; argmax_scratch = L.argmax_scratch; win_scratch = L.win_scratch;
    }
    size_t esize() const { return dtype == TTUP_DTYPE_F32 ? 4 : 2; }
    ~ttup_wasb();
};

namespace ttup {
int run_ops(ttup_wasb* net, int mb, hipStream_t st);
// regions of every op's output that the heatmap rows / columns [lo, hi) depend on (fp32 layer-by-layer graphs only)
int compute_roi(ttup_wasb* net, int lo, int hi, int lo2 = 0, int hi2 = 0);          // [lo2, hi2): the heatmap region of the class-2 samples (conv.h Roi); empty = one class
// certified argmax (csrc/certify.hip)
void cert_free(ttup_wasb* net);
int cert_begin(ttup_wasb* net, int batch, hipStream_t caller);                                   // reset per-call state
int cert_scan(ttup_wasb* net, const float* heat, const long long* argmax, int b0, int mb, hipStream_t st);   // candidates + crop plan of one micro-batch
int cert_finish(ttup_wasb* net, const float* x_dev, const uint8_t* frames_dev, int n_frames, int src_h, int src_w, int batch,
                int64_t* argmax_dev, float* win_dev, hipStream_t caller);                        // fp32 crops, resolve
}  // namespace ttup
