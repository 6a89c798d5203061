// Certified argmax: the production CNN runs in bf16, the reference (balldetection/models/wasb.py, fp32) takes
// `torch.argmax` of the fp32 heatmap (balldetection/helper_balldetection.py:50).  north_star asks for bit-exact argmax
// indices, and bf16 rounding (|error| <= eps, a bound the caller calibrates against the fp32 path) can reorder pixels
// whose fp32 values are closer than 2*eps.  So the index is made exact by construction instead of by luck:
//
//   1. scan      every pixel whose bf16 value is within 2*eps of the bf16 maximum is a candidate -- no other pixel can be
//                the fp32 argmax.  One streaming pass over the fp32-stored heatmap (3.6 MB per frame, HBM-bound).
//   2. plan      one candidate: certified as is.  Several: they are grouped into crops of `Hc x Wc` pixels whose origin is a
//                multiple of 8 (the three stride-2 levels and the nearest-neighbour upsampling then sample exactly as in the
//                full frame).  A heatmap pixel depends on the inputs within R = 72 pixels (measured receptive-field radius 71),
//                so every candidate at least R inside its crop -- or next to a true image border, where the crop's zero padding
//                IS the frame's -- gets bit for bit the value the fp32 path computes on the whole frame.
//   3. crops     the selected windows are pre-processed again in fp32 from the uint8 frames and run through the SAME fp32
//                graph (csrc/conv_f32.hip, the fp32 matrix pipe) as a small batch; everything is sized on the device
//                (`n_active`), nothing synchronises with the host.
//   4. resolve   the candidate with the largest fp32 value (ties -> smaller index, like torch.argmax) becomes the index,
//                and its 3x3 window is taken from the fp32 crop, so the sub-pixel fit also sees fp32 values.
// Heatmaps whose candidates overflow the budget (more than K candidates, more than `maxc` crops, crop list full) are
// flagged 2 in `status`; the caller decides (the Python shim re-runs those frames on the full-frame fp32 handle).
#include "wasb_net.h"
#include <stdlib.h>

namespace ttup {

namespace {

struct Best { float v; long long i; };
__device__ __forceinline__ bool better(float v, long long i, float bv, long long bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

// ---- 1. candidates of each heatmap of a micro-batch: heat (n_maps, hw) fp32, argmax (n_maps)
__global__ __launch_bounds__(256) void cert_scan_kernel(const float* __restrict__ heat, const long long* __restrict__ argmax, long long hw,
                                                        float two_eps, int K, int* __restrict__ cand_idx, int* __restrict__ cand_cnt,
                                                        float* __restrict__ cand_bf, float guard_two_eps, int* __restrict__ guard_cnt) {
    const int map = blockIdx.y;
    const float* h = heat + (size_t)map * hw;
    const float hmax = h[argmax[map]];
    if (hmax != hmax) return;                 // NaN maximum: torch.argmax returns the first NaN, which the bf16 pass already did
    const float thr = hmax - two_eps;
    // guard band: pixels just below the candidate band, down to 2 * (GUARD * eps).  A heatmap without any keeps its candidate set
    // -- and with it its certified result -- when eps is widened by up to the factor GUARD (the shim then re-runs only the others)
    const float gthr = guard_cnt ? hmax - guard_two_eps : thr;
    const long long quads = hw / 4;
    const float4* h4 = (const float4*)h;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const float4 v = h4[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e[k] >= thr) {
                const int slot = atomicAdd(&cand_cnt[map], 1);
                if (slot < K) { cand_idx[(size_t)map * K + slot] = (int)(q * 4 + k); cand_bf[(size_t)map * K + slot] = e[k]; }
            } else if (e[k] >= gthr) atomicAdd(&guard_cnt[map], 1);
    }
}

struct PlanArgs {
    int* cand_idx; int* cand_cnt; int* cand_crop; int* crop_rec; int* n_crops; int* status; unsigned long long* stats; float* cand_bf;
    int K, maxc, max_crops, H, W, Hc, Wc, R, frame0, exact;
    int C, maxf;          // heatmap channels per frame (1 ball, 13 table keypoints) and crops a frame may use in all
    int audit_mod, audit_phase;          // audit crops (ttup_wasb_certify_audit_crops): 0 = off
    int small;                           // class-2 crops (conv.h Roi): valid core positions R + 1 .. R + small of a crop whose candidates all fit there; 0 = off
    const int* guard_cnt;
    float* margin;
};

// valid core of a crop along one axis: positions whose value AND 3x3 neighbourhood are exact
// (small > 0: a class-2 crop, pruned to the cone of the positions R + 1 .. R + small: only those are exact)
__device__ __forceinline__ void core_range(int o, int c, int full, int R, int small, int& lo, int& hi) {
    lo = (o == 0) ? 0 : o + R + 1;
    hi = (o + c == full) ? full : (small > 0 ? o + R + 1 + small : o + c - R - 1);
}

// ---- 2. one workgroup (one wave) per FRAME, its C heatmaps (channels) in turn: the wave sorts a heatmap's candidates by index (the
// scan appends in arbitrary order; a rank sort: every lane counts the smaller indices of its elements), lane 0 assigns them to crops.
// The crops belong to the frame: a crop is one fp32 pass over a window of the frame and yields ALL C channels there, so the
// keypoint heatmaps of the table detector share crops (the ball detector has C = 1).  K is sized for the flat top of a wide
// saturated blob (a few hundred equal pixels inside one crop core), which used to overflow a 32-entry list and send the heatmap
// to the full-frame fp32 path.
constexpr int CERT_MAX_K = 512;
constexpr int CERT_MAX_FRAME_CROPS = 32;
constexpr int CERT_PENDING = 8;          // provisional status of a heatmap whose crops wait for their ids on the shared list
constexpr int CERT_AUDIT_ONLY = 16;      // its crop only MEASURES (audit crop of a single-candidate heatmap): index and window stay the bf16 path's
__global__ __launch_bounds__(64) void cert_plan_kernel(PlanArgs a) {
    // The whole wave works on the assignment (round 4, second half): in its first form lane 0 walked the sorted list in GLOBAL memory --
    // a dependent load per candidate, a read-modify-write per candidate for the final crop ids: 19 us on average and up to 63 us per
    // micro-batch on varied content with one lane; 12 / 26 us in this form.  (The 0.18 ms average / 0.7 ms maximum this kernel shows
    // in the two-lane kernel trace is NOT its own time: its eight small workgroups wait for a slot while the other lane's persistent
    // kernels -- two 256-VGPR waves per SIMD on every CU -- run to their end; the trace counts from the dispatch.)  Same decisions as
    // the serial walk: a candidate goes to the FIRST crop of the frame's list whose core holds it; the first candidate (in index
    // order) that no crop holds opens a new one.
    __shared__ int s_idx[CERT_MAX_K];            // as scanned, then the crop slot of every sorted candidate
    __shared__ float s_bf[CERT_MAX_K];
    __shared__ int s_sorted[CERT_MAX_K];
    __shared__ int my_y0[CERT_MAX_FRAME_CROPS], my_x0[CERT_MAX_FRAME_CROPS], my_small[CERT_MAX_FRAME_CROPS];
    __shared__ int n_my_s, s_base;
    constexpr int PER = CERT_MAX_K / 64;         // candidates per lane
    const int lane = threadIdx.x;
    const int frame = a.frame0 + blockIdx.x;
    if (lane == 0) n_my_s = 0;
    __syncthreads();
    for (int ch = 0; ch < a.C; ++ch) {
        const int map = frame * a.C + ch;
        const int cnt = a.cand_cnt[map];
        const int gbit = a.guard_cnt[map] > 0 ? 4 : 0;          // status bit 2: the guard band is not empty
        if (lane == 0) { atomicAdd(&a.stats[0], 1ull); a.margin[map] = __int_as_float(0x7f800000); }
        // exact-window mode: a single candidate still gets its fp32 crop (the index is certain, the 3x3 window becomes fp32 too).
        // Audit crops: so does the single candidate of ONE channel of every audit_mod-th frame -- its crop reports |bf16 - fp32| at the
        // winner like every crop does (cand_bf / stats[6]), which is how a frame whose error exceeds eps WITHOUT producing a near-tie
        // gets noticed between two strip audits
        const bool audit_pick = a.audit_mod > 0 && cnt == 1 && (frame + a.audit_phase) % a.audit_mod == 0 && ch == (frame / a.audit_mod) % a.C;
        if (cnt <= 0 || (cnt == 1 && !a.exact && !audit_pick)) { if (lane == 0) { a.status[map] = 0 | gbit; atomicAdd(&a.stats[1], 1ull); } continue; }
        if (cnt > a.K) { if (lane == 0) { a.status[map] = 2 | gbit; atomicAdd(&a.stats[3], 1ull); atomicAdd(&a.stats[8], 1ull); } continue; }
        int* ci = a.cand_idx + (size_t)map * a.K;
        float* cb = a.cand_bf + (size_t)map * a.K;
        __syncthreads();                                          // (the previous channel is done with the shared lists)
        for (int i = lane; i < cnt; i += 64) { s_idx[i] = ci[i]; s_bf[i] = cb[i]; }
        __syncthreads();
        for (int i = lane; i < cnt; i += 64) {
            const int v = s_idx[i];
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += s_idx[j] < v;          // pixel indices are distinct: ranks are a permutation
            ci[rank] = v; cb[rank] = s_bf[i]; s_sorted[rank] = v;
        }
        __syncthreads();
        if (lane == 0 && cnt == 1) atomicAdd(&a.stats[7], 1ull);
        // the lane's candidates lane, lane + 64, ...: position and the slot of the first crop that holds them (-1: none yet)
        int cy[PER], cx[PER], found[PER];
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int i = lane + 64 * m;
            const int v = i < cnt ? s_sorted[i] : 0;
            cy[m] = v / a.W; cx[m] = v - cy[m] * a.W; found[m] = -1;
        }
        // crops the frame has so far (from its earlier channels) are tried first; new ones are added behind them and dropped again
        // if this heatmap turns out to need more than its budget
        int n_my = n_my_s;
        const int n_before = n_my;
        bool over = false;
        int c_from = 0;                                           // crops [c_from, n_my) have not been tried on the uncovered candidates yet
        while (true) {
            for (int c = c_from; c < n_my; ++c) {
                int ylo, yhi, xlo, xhi;
                core_range(my_y0[c], a.Hc, a.H, a.R, my_small[c], ylo, yhi);
                core_range(my_x0[c], a.Wc, a.W, a.R, my_small[c], xlo, xhi);
#pragma unroll
                for (int m = 0; m < PER; ++m)
                    if (found[m] < 0 && cy[m] >= ylo && cy[m] < yhi && cx[m] >= xlo && cx[m] < xhi) found[m] = c;
            }
            c_from = n_my;
            int first = -1;                                       // the first candidate (index order) that no crop holds
#pragma unroll
            for (int m = 0; m < PER; ++m) {
                const unsigned long long unc = __builtin_amdgcn_ballot_w64(lane + 64 * m < cnt && found[m] < 0);
                if (first < 0 && unc) first = 64 * m + __builtin_ctzll(unc);
            }
            if (first < 0) break;
            if (n_my - n_before >= a.maxc || n_my >= a.maxf) { over = true; break; }
            if (lane == 0) {
                // A new crop, centred on the bounding box of the candidates from `first` on that can share it (it is the top-most
                // uncovered one: the list is sorted by index).  With the origin ROUNDED to a multiple of 8 the core covers centre - 7 ..
                // centre + 7 at least, so a cluster of up to 15 x 15 pixels -- the flat top of a saturated blob -- takes ONE crop (a crop
                // centred on the first candidate, the top row of the blob, left its lower half to a second crop).
                const int k = first, fy = s_sorted[k] / a.W, fx = s_sorted[k] % a.W;
                const int span_y = a.Hc - 2 * a.R - 2 - 7, span_x = a.Wc - 2 * a.R - 2 - 7;
                int ylo = fy, yhi = fy, xlo = fx, xhi = fx;
                for (int j = k + 1; j < cnt; ++j) {
                    const int yj = s_sorted[j] / a.W, xj = s_sorted[j] % a.W;
                    if (yj - fy >= span_y) break;
                    const int nxlo = xj < xlo ? xj : xlo, nxhi = xj > xhi ? xj : xhi;
                    if (nxhi - nxlo >= span_x) continue;
                    xlo = nxlo; xhi = nxhi; yhi = yj;
                }
                // Class 2 (round 6): a cluster that fits a core of a.small positions with the same rounding slack (span <= small - 8)
                // -- every single candidate does -- is centred on the core R + 1 .. R + small instead of the crop's centre; its fp32 pass
                // is pruned to the cone of THAT core, which ends 8 pixels short of the crop's last row / column (conv.h Roi).
                const int sm = (a.small > 0 && yhi - ylo <= a.small - 8 && xhi - xlo <= a.small - 8) ? a.small : 0;
                const int mid = sm ? a.R + 1 + sm / 2 : -1;          // crop position of the cluster's centre (-1: the crop's own centre)
                auto origin = [mid](int c, int crop, int full) {
                    int o = ((c - (mid < 0 ? crop / 2 : mid) + 4) >> 3) << 3;
                    return o < 0 ? 0 : (o > full - crop ? full - crop : o);
                };
                int y0 = origin((ylo + yhi) / 2, a.Hc, a.H), x0 = origin((xlo + xhi) / 2, a.Wc, a.W);
                {
                    int cylo, cyhi, cxlo, cxhi;
                    core_range(y0, a.Hc, a.H, a.R, sm, cylo, cyhi);
                    core_range(x0, a.Wc, a.W, a.R, sm, cxlo, cxhi);
                    if (!(fy >= cylo && fy < cyhi && fx >= cxlo && fx < cxhi)) { y0 = origin(fy, a.Hc, a.H); x0 = origin(fx, a.Wc, a.W); }      // (cannot happen for spans < 15 / small - 7; kept as a guard)
                }
                my_y0[n_my] = y0; my_x0[n_my] = x0; my_small[n_my] = sm;
            }
            ++n_my;
            __syncthreads();
        }
        if (over) { if (lane == 0) { a.status[map] = 2 | gbit; atomicAdd(&a.stats[3], 1ull); atomicAdd(&a.stats[9], 1ull); } continue; }          // (n_my_s keeps the list without this heatmap's new crops)
#pragma unroll
        for (int m = 0; m < PER; ++m)
            if (lane + 64 * m < cnt) a.cand_crop[(size_t)map * a.K + lane + 64 * m] = found[m];          // slot in the frame's list for now
        __syncthreads();                                          // (every lane has read n_my_s)
        if (lane == 0) { n_my_s = n_my; a.status[map] = CERT_PENDING | gbit | ((audit_pick && !a.exact) ? CERT_AUDIT_ONLY : 0); }
    }
    __syncthreads();
    const int n_my = n_my_s;
    if (n_my == 0) return;
    if (lane == 0) {
        const int base = atomicAdd(a.n_crops, n_my);
        s_base = base;
        for (int c = 0; c < n_my && base + c < a.max_crops; ++c) {          // (records also for a list that fills up half way: the slots are run)
            int* rec = a.crop_rec + 4 * (base + c);
            rec[0] = frame; rec[1] = my_y0[c]; rec[2] = my_x0[c]; rec[3] = my_small[c] > 0 ? 1 : 0;
        }
        if (!(base + n_my > a.max_crops)) atomicAdd(&a.stats[4], (unsigned long long)n_my);
        if (!(base + n_my > a.max_crops)) {
            int ns = 0;
            for (int c = 0; c < n_my; ++c) ns += my_small[c] > 0;
            if (ns) atomicAdd(&a.stats[11], (unsigned long long)ns);
        }
    }
    __threadfence_block();
    __syncthreads();
    const int base = s_base;
    const bool full = base + n_my > a.max_crops;                          // crop list full: the frame's heatmaps stay uncertified
    for (int ch = 0; ch < a.C; ++ch) {
        const int map = frame * a.C + ch;
        const int st = a.status[map];          // (written by lane 0 of this workgroup above: visible after the barrier)
        if (!(st & CERT_PENDING)) continue;
        const int gbit = st & 4, abit = st & CERT_AUDIT_ONLY;
        if (full && abit) { if (lane == 0) { a.status[map] = 0 | gbit; atomicAdd(&a.stats[1], 1ull); } continue; }          // (no room for the audit: still a certified single candidate)
        if (full) { if (lane == 0) { a.status[map] = 2 | gbit; atomicAdd(&a.stats[3], 1ull); atomicAdd(&a.stats[10], 1ull); } continue; }
        const int cnt = a.cand_cnt[map];
        for (int k = lane; k < cnt; k += 64) a.cand_crop[(size_t)map * a.K + k] += base;
        if (lane == 0) {
            a.status[map] = 1 | gbit | abit;
            atomicAdd(&a.stats[2], 1ull);
            atomicAdd(&a.stats[5], (unsigned long long)cnt);
        }
    }
}

__global__ void cert_active_kernel(const int* n_crops, int* n_active, int CH, int nchunks, int max_crops, const int* crop_rec, int* roi_flag,
                                   int H, int W, int Hc, int Wc) {
    const int c = threadIdx.x;
    if (c >= nchunks) return;
    int n = *n_crops;
    n = n > max_crops ? max_crops : n;
    int v = n - c * CH;
    v = v < 0 ? 0 : (v > CH ? CH : v);
    n_active[c] = v;
    // cone pruning applies to INTERIOR crops: a crop on an image border has a core that reaches that border (its zero padding IS the
    // frame's), i.e. a wider cone: those are computed in full (one flag per crop; the kernels skip the tiles / pixels outside an op's
    // region for the flagged samples only)
    for (int j = 0; j < v; ++j) {
        const int* rec = crop_rec + 4 * (c * CH + j);
        roi_flag[c * CH + j] = (rec[1] <= 0 || rec[2] <= 0 || rec[1] + Hc >= H || rec[2] + Wc >= W) ? 0 : (rec[3] ? 2 : 1);
    }
}

// crop windows of a caller-supplied fp32 NCHW input (the `forward(x)` entry): -> fp32 NHWC16
__global__ void cert_gather_kernel(const float* __restrict__ x, int in_ch, int H, int W, const int* __restrict__ crops, int crop0,
                                   const int* __restrict__ n_active, int Hc, int Wc, float* __restrict__ out, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cx = (int)(i % Wc);
    long long p = i / Wc;
    const int cy = (int)(p % Hc);
    const int j = (int)(p / Hc);
    if (j >= *n_active) return;
    const int* rec = crops + 4 * (crop0 + j);
    const size_t hw = (size_t)H * W, pix = (size_t)(rec[1] + cy) * W + rec[2] + cx;
    float* o = out + (size_t)i * 16;
    for (int c = 0; c < 16; ++c) o[c] = c < in_ch ? x[((size_t)rec[0] * in_ch + c) * hw + pix] : 0.f;
}

// ---- 4a. fp32 value and 3x3 window of every candidate whose crop is in this chunk
__global__ void cert_lookup_kernel(const int* __restrict__ cand_idx, const int* __restrict__ cand_cnt, const int* __restrict__ cand_crop,
                                   const int* __restrict__ status, const int* __restrict__ crop_rec, const float* __restrict__ crop_heat,
                                   int K, int H, int W, int Hc, int Wc, int crop0, int CH, int n_maps, float* __restrict__ cand_val, float* __restrict__ cand_win,
                                   const float* __restrict__ cand_bf, unsigned long long* __restrict__ stats, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_maps * K) return;
    const int map = i / K, k = i % K;
    if ((status[map] & 3) != 1 || k >= cand_cnt[map]) return;
    const int id = cand_crop[i];
    if (id < crop0 || id >= crop0 + CH) return;
    const int* rec = crop_rec + 4 * id;
    const int gy = cand_idx[i] / W, gx = cand_idx[i] % W;
    const float* h = crop_heat + ((size_t)(id - crop0) * C + map % C) * Hc * Wc;          // crop_heat (CH, C, Hc, Wc)
    const float vf = h[(size_t)(gy - rec[1]) * Wc + (gx - rec[2])];
    cand_val[i] = vf;
    // audit of the error bound: |bf16 - fp32| at every candidate comes for free here; the running maximum sits in stats[6] (the
    // bits of a non-negative float order like the unsigned integer they spell)
    const float err = fabsf(cand_bf[i] - vf);
    if (err == err) atomicMax(&stats[6], (unsigned long long)__float_as_uint(err));
    for (int t = 0; t < 9; ++t) {
        const int y = gy + t / 3 - 1, x = gx + t % 3 - 1;
        float v = 0.f;                        // zero padding outside the IMAGE (helper_balldetection.py:55-64)
        if (y >= 0 && y < H && x >= 0 && x < W) v = h[(size_t)(y - rec[1]) * Wc + (x - rec[2])];
        cand_win[(size_t)i * 9 + t] = v;
    }
}

// ---- 4b. the fp32 winner of every heatmap that needed crops
__global__ void cert_resolve_kernel(const int* __restrict__ cand_idx, const int* __restrict__ cand_cnt, int* __restrict__ status,
                                    const float* __restrict__ cand_val, const float* __restrict__ cand_win, int K, int n_maps,
                                    long long* __restrict__ argmax, float* __restrict__ win, float* __restrict__ margin) {
    const int map = blockIdx.x * blockDim.x + threadIdx.x;
    if (map >= n_maps || (status[map] & 3) != 1) return;
    // an audit crop has left its |bf16 - fp32| in stats[6] (cert_lookup_kernel): the heatmap's result is what the bf16 path returned,
    // whatever frames an audit happens to look at (outputs do not depend on the audit phase); status back to "single candidate"
    if (status[map] & CERT_AUDIT_ONLY) { status[map] = status[map] & 4; return; }
    const int cnt = cand_cnt[map];
    float bv = cand_val[(size_t)map * K];
    long long bi = cand_idx[(size_t)map * K];
    int bk = 0;
    float second = -__int_as_float(0x7f800000);
    for (int k = 1; k < cnt; ++k) {
        const float v = cand_val[(size_t)map * K + k];
        const long long i = cand_idx[(size_t)map * K + k];
        if (better(v, i, bv, bi)) { second = bv; bv = v; bi = i; bk = k; }
        else if (v > second) second = v;
    }
    argmax[map] = bi;
    margin[map] = bv - second;          // how far the fp32 winner is ahead of the best other candidate (measurement: "reference-ambiguous" share)
    for (int t = 0; t < 9; ++t) win[(size_t)map * 9 + t] = cand_win[((size_t)map * K + bk) * 9 + t];
}

}  // namespace

void cert_free(ttup_wasb* net) {
    CertState& c = net->cert;
    if (c.cropnet) { ttup_wasb_destroy(c.cropnet); c.cropnet = nullptr; }
    for (auto& sl : c.slot) {
        void* ptrs[] = {sl.cand_idx, sl.cand_cnt, sl.cand_crop, sl.cand_val, sl.cand_win, sl.cand_bf, sl.crop_rec, sl.n_crops, sl.n_active, sl.status, sl.guard_cnt, sl.margin, sl.roi_flag};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.read_status) (void)hipEventDestroy(sl.read_status);
        if (sl.read_info) (void)hipEventDestroy(sl.read_info);
        if (sl.read_margin) (void)hipEventDestroy(sl.read_margin);
        sl = CertState::Slot();
    }
    if (c.stats) (void)hipFree(c.stats);
    if (c.crop_heat) (void)hipFree(c.crop_heat);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    if (c.lanes_done) (void)hipEventDestroy(c.lanes_done);
    c.stats = nullptr; c.crop_heat = nullptr; c.stream = nullptr; c.lanes_done = nullptr;
    c.enabled = false;
}

int cert_begin(ttup_wasb* net, int batch, hipStream_t caller) {
    CertState& c = net->cert;
    c.cur ^= 1;
    CertState::Slot& sl = c.slot[c.cur];
    TTUP_HIP_CHECK(hipStreamWaitEvent(caller, sl.done, 0));          // the call that last used this slot has finished its fp32 passes
    TTUP_HIP_CHECK(hipStreamWaitEvent(caller, sl.read_status, 0));   // ... and its caller's status / info copies have been made
    TTUP_HIP_CHECK(hipStreamWaitEvent(caller, sl.read_info, 0));
    TTUP_HIP_CHECK(hipStreamWaitEvent(caller, sl.read_margin, 0));   // (own event: hipEventRecord overwrites, and the margin copy may be issued on another stream than the status copy)
    const size_t maps = (size_t)batch * net->n_out;
    TTUP_HIP_CHECK(hipMemsetAsync(sl.cand_cnt, 0, maps * sizeof(int), caller));
    TTUP_HIP_CHECK(hipMemsetAsync(sl.guard_cnt, 0, maps * sizeof(int), caller));
    TTUP_HIP_CHECK(hipMemsetAsync(sl.n_crops, 0, sizeof(int), caller));
    TTUP_HIP_CHECK(hipMemsetAsync(sl.status, 0, maps * sizeof(int), caller));
    return TTUP_OK;
}

int cert_scan(ttup_wasb* net, const float* heat, const long long* argmax, int b0, int mb, hipStream_t st) {
    CertState& c = net->cert;
    CertState::Slot& sl = c.slot[c.cur];
    const long long hw = (long long)net->H * net->W;
    int nblk = (int)(hw / 4 / 256 / 8);           // 8 float4 per thread
    nblk = nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk);
    const int C = net->n_out;          // heat: (mb, C, H, W) -- map index = frame * C + channel
    const size_t m0 = (size_t)b0 * C;
    hipLaunchKernelGGL(cert_scan_kernel, dim3(nblk, mb * C), dim3(256), 0, st, heat, argmax, hw, 2.f * c.eps, c.K,
                       sl.cand_idx + m0 * c.K, sl.cand_cnt + m0, sl.cand_bf + m0 * c.K, 2.f * c.eps * CertState::GUARD, sl.guard_cnt + m0);
    TTUP_LAUNCH_CHECK();
    PlanArgs a;
    a.cand_idx = sl.cand_idx; a.cand_cnt = sl.cand_cnt; a.cand_crop = sl.cand_crop; a.crop_rec = sl.crop_rec; a.n_crops = sl.n_crops;
    a.status = sl.status; a.stats = c.stats; a.cand_bf = sl.cand_bf; a.K = c.K; a.maxc = c.maxc; a.max_crops = c.budget;
    a.H = net->H; a.W = net->W; a.Hc = c.Hc; a.Wc = c.Wc; a.R = c.R; a.frame0 = b0; a.C = C; a.maxf = c.maxf; a.exact = c.exact_windows ? 1 : 0; a.guard_cnt = sl.guard_cnt; a.margin = sl.margin;
    a.audit_mod = c.audit_mod; a.audit_phase = c.audit_phase;
    a.small = c.small;
    hipLaunchKernelGGL(cert_plan_kernel, dim3(mb), dim3(64), 0, st, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

int cert_finish(ttup_wasb* net, const float* x_dev, const uint8_t* frames_dev, int n_frames, int src_h, int src_w, int batch,
                int64_t* argmax_dev, float* win_dev, hipStream_t caller) {
    CertState& c = net->cert;
    CertState::Slot& sl = c.slot[c.cur];
    ttup_wasb* cn = c.cropnet;
    // the fp32 passes run on the handle's own stream, behind everything the caller's stream has seen (the bf16 micro-batches
    // have been joined into it): a following call issued on ANOTHER caller stream overlaps with them
    hipStream_t st = c.stream;
    TTUP_HIP_CHECK(hipEventRecord(c.lanes_done, caller));
    TTUP_HIP_CHECK(hipStreamWaitEvent(st, c.lanes_done, 0));
    hipLaunchKernelGGL(cert_active_kernel, dim3(1), dim3(64), 0, st, sl.n_crops, sl.n_active, c.CH, c.nchunks, c.budget, (const int*)sl.crop_rec, sl.roi_flag,
                       net->H, net->W, c.Hc, c.Wc);
    TTUP_LAUNCH_CHECK();
    // fp32 passes that can hold crops of THIS call: at most maxc per heatmap, at most the caller's budget
    const int C = net->n_out;
    int nch = cdiv(batch * c.maxf < c.budget ? batch * c.maxf : c.budget, c.CH);
    nch = nch > c.nchunks ? c.nchunks : nch;
    for (int ch = 0; ch < nch; ++ch) {
        const int crop0 = ch * c.CH;
        const int* na = sl.n_active + ch;
        cn->use_lane(0);
        float* xin = (float*)cn->tensors[cn->t_input].ptr;
        if (frames_dev) {
            const int rc = launch_preprocess_crops(frames_dev, n_frames, src_h, src_w, net->H, net->W, xin, sl.crop_rec, crop0, na, c.CH, c.Hc, c.Wc, net->in_ch / 3, st);
            if (rc) return rc;
        } else {
            const long long total = (long long)c.CH * c.Hc * c.Wc;
            hipLaunchKernelGGL(cert_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x_dev, net->in_ch, net->H, net->W,
                               (const int*)sl.crop_rec, crop0, na, c.Hc, c.Wc, xin, total);
            TTUP_LAUNCH_CHECK();
        }
        cn->n_active = na;
        cn->roi_flag = cn->op_roi.empty() ? nullptr : sl.roi_flag + crop0;
        Roi hr = cn->out_roi; hr.flag = cn->roi_flag;
        int rc = run_ops(cn, c.CH, st);
        if (rc == TTUP_OK) rc = launch_head(cn->tensors[cn->t_out].ptr, cn->head_w_dev, cn->head_b_dev, C, c.crop_heat, c.CH, c.Hc, c.Wc, 16, TTUP_DTYPE_F32, st, na, &hr);
        cn->n_active = nullptr; cn->roi_flag = nullptr;
        if (rc) return rc;
        const int nthr = batch * C * c.K;
        hipLaunchKernelGGL(cert_lookup_kernel, dim3(cdiv(nthr, 256)), dim3(256), 0, st, (const int*)sl.cand_idx, (const int*)sl.cand_cnt, (const int*)sl.cand_crop,
                           (const int*)sl.status, (const int*)sl.crop_rec, (const float*)c.crop_heat, c.K, net->H, net->W, c.Hc, c.Wc, crop0, c.CH, batch * C,
                           sl.cand_val, sl.cand_win, (const float*)sl.cand_bf, c.stats, C);
        TTUP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(cert_resolve_kernel, dim3(cdiv(batch * C, 64)), dim3(64), 0, st, (const int*)sl.cand_idx, (const int*)sl.cand_cnt, sl.status,
                       (const float*)sl.cand_val, (const float*)sl.cand_win, c.K, batch * C, (long long*)argmax_dev, win_dev, sl.margin);
    TTUP_LAUNCH_CHECK();
    TTUP_HIP_CHECK(hipEventRecord(sl.done, st));
    TTUP_HIP_CHECK(hipStreamWaitEvent(caller, sl.done, 0));          // outputs are final in the caller's stream order
    return TTUP_OK;
}

}  // namespace ttup

using namespace ttup;

int ttup_wasb_create_internal(const void* blob, size_t blob_bytes, int height, int width, int max_batch, int dtype, int micro, int lanes, ttup_wasb** out);

extern "C" int ttup_wasb_set_certify(ttup_wasb* net, float eps_abs, int crop, int max_crops_per_map) {
    TTUP_REQUIRE(net, TTUP_EINVAL, "ttup_wasb_set_certify: null handle");
    if (eps_abs < 0.f) { (void)hipDeviceSynchronize(); cert_free(net); return TTUP_OK; }
    TTUP_REQUIRE(net->dtype == TTUP_DTYPE_BF16, TTUP_EINVAL,
                 "ttup_wasb_set_certify: the certified argmax applies to bf16 handles (an fp32 handle's argmax is the fp32 argmax)");
    TTUP_REQUIRE(eps_abs == eps_abs && crop >= 0 && max_crops_per_map >= 0 && max_crops_per_map <= CERT_MAX_FRAME_CROPS, TTUP_EINVAL, "ttup_wasb_set_certify: bad argument");
    CertState& c = net->cert;
    if (c.enabled) { c.eps = eps_abs; if (crop == 0 && max_crops_per_map == 0) return TTUP_OK; }
    (void)hipDeviceSynchronize();
    cert_free(net);
    c.eps = eps_abs;
    c.small = 0;
    static const int env_maxc = getenv("TTUP_CERT_MAXC") ? atoi(getenv("TTUP_CERT_MAXC")) : 0, env_list = getenv("TTUP_CERT_LIST") ? atoi(getenv("TTUP_CERT_LIST")) : 0;
    c.maxc = max_crops_per_map > 0 ? max_crops_per_map : (env_maxc > 0 && env_maxc <= CERT_MAX_FRAME_CROPS ? env_maxc : 8);
    c.maxf = c.maxc * net->n_out < CERT_MAX_FRAME_CROPS ? c.maxc * net->n_out : CERT_MAX_FRAME_CROPS;      // the channels of a frame share its crops
    // Crop side: 2 R + the core.  The origin of a crop is a multiple of 8 (the 1/8-resolution branch), so a crop centred on a candidate
    // has it within 4 pixels of its centre: the core must hold 8 positions + the 3x3 window = 2 R + 16 at least.
    static const int env_crop = getenv("TTUP_CERT_CROP") ? atoi(getenv("TTUP_CERT_CROP")) : 0;          // experiment knob, read once
    int side = crop > 0 ? crop : (env_crop > 0 ? env_crop : 168);
    TTUP_REQUIRE(side % 8 == 0 && side >= 2 * c.R + 16, TTUP_EINVAL, "ttup_wasb_set_certify: crop %d must be a multiple of 8 and at least %d", side, 2 * c.R + 16);
    c.Hc = side < net->H ? side : net->H;
    c.Wc = side < net->W ? side : net->W;
    // the scan reads float4 quads of whole heatmaps; a crop is exact only when its (clamped) origin is a multiple of 8
    TTUP_REQUIRE(((long long)net->H * net->W) % 4 == 0 && (net->H - c.Hc) % 8 == 0 && (net->W - c.Wc) % 8 == 0, TTUP_EINVAL,
                 "ttup_wasb_set_certify: %dx%d heatmaps with %dx%d crops cannot be certified (H*W %% 4, (H-Hc) %% 8, (W-Wc) %% 8 must be 0)", net->H, net->W, c.Hc, c.Wc);
    static const int env_ch = getenv("TTUP_CERT_CH") ? atoi(getenv("TTUP_CERT_CH")) : 0;          // crops per fp32 pass (experiment knob, read once)
    const int ch_cap = env_ch >= 8 && env_ch <= 512 ? env_ch : 128;          // 128 against 64: varied content +1.7 %, parity mode and noise weights +1.1 % (fewer, fuller passes; round 6)
    c.CH = net->max_batch < ch_cap ? net->max_batch : ch_cap;
    const int per_map = env_list > 0 && env_list <= 16 ? env_list : 4;
    c.max_crops = per_map * net->max_batch > c.CH ? per_map * net->max_batch : c.CH;   // capacity of the call's crop list: four per heatmap on average; the overflow is flagged
    if (c.max_crops < c.maxf) c.max_crops = c.maxf;          // ... and never less than ONE frame may ask for (one-sample handles: re-certification of single frames, round-4 advisor)
    // (round 4: 2 -> 4 and 4 -> 8 crops per heatmap: on pure noise weights 5.5 % of the heatmaps overflowed and went to the full-frame fp32 path --
    // 3.6 ms each, the price of 32 crops; now none: 737 -> 854 frames/s, tools/noise_regime.py)
    c.nchunks = cdiv(c.max_crops, c.CH);
    TTUP_REQUIRE(c.K <= CERT_MAX_K, TTUP_EINVAL, "ttup_wasb_set_certify: candidate list %d longer than the plan kernel's %d", c.K, CERT_MAX_K);
    TTUP_REQUIRE(c.nchunks <= 64, TTUP_EINVAL, "ttup_wasb_set_certify: max_batch %d too large", net->max_batch);
    c.max_crops = c.nchunks * c.CH;
    c.budget = net->max_batch < c.max_crops ? (net->max_batch > c.CH ? net->max_batch : c.CH) : c.max_crops;      // default: one crop per heatmap
    const size_t nb = (size_t)net->max_batch * net->n_out;          // heatmaps per call
    for (auto& sl : c.slot) {
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_idx, nb * c.K * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_cnt, nb * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_crop, nb * c.K * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_val, nb * c.K * sizeof(float)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_bf, nb * c.K * sizeof(float)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.guard_cnt, nb * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.cand_win, nb * c.K * 9 * sizeof(float)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.crop_rec, (size_t)c.max_crops * 4 * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.n_crops, sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.n_active, (size_t)c.nchunks * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.roi_flag, (size_t)c.max_crops * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.status, nb * sizeof(int)));
        TTUP_HIP_CHECK(hipMalloc((void**)&sl.margin, nb * sizeof(float)));
        TTUP_HIP_CHECK(hipMemsetD32((hipDeviceptr_t)sl.margin, 0x7f800000, nb));          // +inf: heatmaps never planned or resolved (ttup.h)
        TTUP_HIP_CHECK(hipMemset(sl.status, 0, nb * sizeof(int)));
        TTUP_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        TTUP_HIP_CHECK(hipEventCreateWithFlags(&sl.read_status, hipEventDisableTiming));
        TTUP_HIP_CHECK(hipEventCreateWithFlags(&sl.read_info, hipEventDisableTiming));
        TTUP_HIP_CHECK(hipEventCreateWithFlags(&sl.read_margin, hipEventDisableTiming));
    }
    TTUP_HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    TTUP_HIP_CHECK(hipEventCreateWithFlags(&c.lanes_done, hipEventDisableTiming));
    TTUP_HIP_CHECK(hipMalloc((void**)&c.stats, 12 * sizeof(unsigned long long)));
    TTUP_HIP_CHECK(hipMemset(c.stats, 0, 12 * sizeof(unsigned long long)));
    TTUP_HIP_CHECK(hipMalloc((void**)&c.crop_heat, (size_t)c.CH * net->n_out * c.Hc * c.Wc * sizeof(float)));
    const int rc = ttup_wasb_create_internal(net->blob.data(), net->blob.size(), c.Hc, c.Wc, c.CH, TTUP_DTYPE_F32, c.CH, 1, &c.cropnet);
    if (rc) { cert_free(net); return rc; }
    // cone pruning of the crop net: an interior crop's candidates lie R + 1 pixels inside it, their 3x3 windows one more: only the
    // heatmap rows / columns [R, side - R) are ever read (lookup kernel), and every layer only has to produce what those depend on
    static const bool no_cone = getenv("TTUP_NO_CONE") != nullptr || getenv("TTUP_F32_EXACT") != nullptr || getenv("TTUP_F32_DIRECT") != nullptr;
    if (!no_cone && c.Hc == c.Wc && c.Hc > 2 * c.R + 2 && c.Hc < net->H && c.Wc < net->W) {
        // class 2: a 16-pixel heatmap region (14 candidate positions + their 3x3 windows) at the crop's corner-aligned end of the core
        // range -- its cone is the crop's first 160 rows / columns, i.e. one 16-pixel tile row / column less in the full-resolution layers
        static const bool no_small = getenv("TTUP_CERT_SMALL") != nullptr && atoi(getenv("TTUP_CERT_SMALL")) == 0;
        c.small = (!no_small && c.Hc >= 2 * c.R + 24) ? 14 : 0;
        const int rc2 = compute_roi(c.cropnet, c.R, c.Hc - c.R, c.R, c.small ? c.R + c.small + 2 : c.R);
        if (rc2) { cert_free(net); return rc2; }
    }
    c.enabled = true;
    return TTUP_OK;
}

extern "C" int ttup_wasb_certify_budget(ttup_wasb* net, int max_crops) {
    TTUP_REQUIRE(net && net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_budget: the certified argmax is not enabled on this handle");
    TTUP_REQUIRE(max_crops >= 1, TTUP_EINVAL, "ttup_wasb_certify_budget: budget must be positive");
    net->cert.budget = max_crops < net->cert.max_crops ? max_crops : net->cert.max_crops;
    return TTUP_OK;
}

// The scan on its own (measurement aid and test hook): candidates of n_maps fp32 heatmaps within 2*eps_abs of each map's value at
// argmax_dev[map].  cand_cnt_dev must be zeroed by the caller; cand_idx_dev / cand_bf_dev hold K entries per map.
extern "C" int ttup_certify_scan(const float* heat_dev, const int64_t* argmax_dev, int n_maps, int height, int width, float eps_abs, int K,
                                 int* cand_idx_dev, int* cand_cnt_dev, float* cand_bf_dev, void* stream) {
    TTUP_REQUIRE(heat_dev && argmax_dev && cand_idx_dev && cand_cnt_dev && cand_bf_dev, TTUP_EINVAL, "ttup_certify_scan: null pointer");
    TTUP_REQUIRE(n_maps >= 0 && height > 0 && width > 0 && ((long long)height * width) % 4 == 0 && K > 0 && eps_abs >= 0.f, TTUP_EINVAL, "ttup_certify_scan: bad argument");
    if (n_maps == 0) return TTUP_OK;
    const long long hw = (long long)height * width;
    int nblk = (int)(hw / 4 / 256 / 8);
    nblk = nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk);
    hipLaunchKernelGGL(cert_scan_kernel, dim3(nblk, n_maps), dim3(256), 0, (hipStream_t)stream, heat_dev, (const long long*)argmax_dev, hw, 2.f * eps_abs, K,
                       cand_idx_dev, cand_cnt_dev, cand_bf_dev, 0.f, (int*)nullptr);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

// The audit's error measure, max |a - b|, in one pass and without torch's element-wise kernels (the audits run on a side stream beside
// the CNN: no packed fp32 here, common.h).  NaN anywhere gives NaN (its bit pattern orders above +inf).  The 2-D form compares the
// columns [c0, c1) of `rows` rows of `width` floats (a strip audit leaves out the columns whose receptive field reaches the strip's
// artificial zero padding) and can keep a running maximum in `out`.
TTUP_NO_PACKED_FP32_BEGIN          // (bracketed kernels call builtins only: a HIP header function would stay an out-of-line call, no_packed_fp32_begin.h)
namespace ttup { namespace {
__global__ __launch_bounds__(256) void max_abs_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long rows, int width, int c0, int ncol,
                                                           unsigned* __restrict__ out) {
    float m = 0.f;
    bool nan = false;
    const long long n = rows * ncol;
    for (long long i = (long long)ttup_bid_x() * 256 + ttup_tid_x(); i < n; i += (long long)ttup_gsize_x()) {
        const long long e = ncol == width ? i : (i / ncol) * width + c0 + i % ncol;
        const float d = fabsf(a[e] - b[e]);
        nan |= d != d;
        m = d > m ? d : m;
    }
    unsigned bits = nan ? 0x7fc00000u : __builtin_bit_cast(unsigned, m);            // non-negative floats order like their bit patterns
#pragma unroll
    for (int k = 0; k < 6; ++k) { const unsigned o = (unsigned)__builtin_amdgcn_ds_bpermute(((ttup_tid_x() & 63) ^ (32 >> k)) << 2, (int)bits); bits = o > bits ? o : bits; }
    if ((ttup_tid_x() & 63) == 0 && bits) __hip_atomic_fetch_max(out, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// dst[r][j] = src[r][x0 + j]: a column strip of (rows, width) floats (the audit's strip of the pre-processed input)
__global__ __launch_bounds__(256) void slice_columns_kernel(const float* __restrict__ src, long long rows, int width, int x0, int w, float* __restrict__ dst) {
    const long long n = rows * w;
    for (long long i = (long long)ttup_bid_x() * 256 + ttup_tid_x(); i < n; i += (long long)ttup_gsize_x()) dst[i] = src[(i / w) * width + x0 + i % w];
}
} }
TTUP_NO_PACKED_FP32_END

extern "C" int ttup_max_abs_diff(const float* a_dev, const float* b_dev, long long n, float* out_dev, void* stream) {
    return ttup_max_abs_diff_cols(a_dev, b_dev, 1, n, 0, n, out_dev, 0, stream);
}

extern "C" int ttup_max_abs_diff_cols(const float* a_dev, const float* b_dev, long long rows, long long width, long long c0, long long c1, float* out_dev,
                                      int accumulate, void* stream) {
    TTUP_REQUIRE(a_dev && b_dev && out_dev && rows >= 0 && width >= 0 && c0 >= 0 && c0 <= c1 && c1 <= width, TTUP_EINVAL, "ttup_max_abs_diff: bad argument");
    TTUP_REQUIRE(rows <= 1 || width < (1ll << 31), TTUP_EINVAL, "ttup_max_abs_diff_cols: rows wider than 2^31 floats");
    if (!accumulate) TTUP_HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(float), (hipStream_t)stream));
    const long long n = rows * (c1 - c0);
    if (n == 0) return TTUP_OK;
    long long nblk = (n + 256 * 16 - 1) / (256 * 16);
    nblk = nblk > 2048 ? 2048 : nblk;
    if (rows == 1) {          // one row: a flat range (any length)
        hipLaunchKernelGGL(max_abs_diff_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a_dev + c0, b_dev + c0, c1 - c0, 1, 0, 1, (unsigned*)out_dev);
    } else {
        hipLaunchKernelGGL(max_abs_diff_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a_dev, b_dev, rows, (int)width, (int)c0, (int)(c1 - c0), (unsigned*)out_dev);
    }
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

extern "C" int ttup_slice_columns(const float* src_dev, long long rows, int width, int x0, int w, float* dst_dev, void* stream) {
    TTUP_REQUIRE(src_dev && dst_dev && rows >= 0 && width > 0 && x0 >= 0 && w >= 0 && x0 + w <= width, TTUP_EINVAL, "ttup_slice_columns: bad argument");
    const long long n = rows * w;
    if (n == 0) return TTUP_OK;
    long long nblk = (n + 256 * 8 - 1) / (256 * 8);
    nblk = nblk > 4096 ? 4096 : nblk;
    hipLaunchKernelGGL(slice_columns_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, src_dev, rows, width, x0, w, dst_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

extern "C" int ttup_wasb_certify_exact_windows(ttup_wasb* net, int on) {
    TTUP_REQUIRE(net && net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_exact_windows: the certified argmax is not enabled on this handle");
    net->cert.exact_windows = on != 0;
    return TTUP_OK;
}

extern "C" int ttup_wasb_certify_audit_crops(ttup_wasb* net, int every, int phase) {
    TTUP_REQUIRE(net && net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_audit_crops: the certified argmax is not enabled on this handle");
    TTUP_REQUIRE(every >= 0 && phase >= 0, TTUP_EINVAL, "ttup_wasb_certify_audit_crops: every and phase must not be negative");
    net->cert.audit_mod = every;
    net->cert.audit_phase = every > 0 ? phase % every : 0;
    return TTUP_OK;
}

extern "C" int ttup_wasb_certify_info(ttup_wasb* net, int* info_dev, void* stream) {
    TTUP_REQUIRE(net && info_dev, TTUP_EINVAL, "ttup_wasb_certify_info: null pointer");
    TTUP_REQUIRE(net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_info: the certified argmax is not enabled on this handle");
    CertState::Slot& sl = net->cert.slot[net->cert.cur];
    TTUP_HIP_CHECK(hipMemcpyAsync(info_dev, sl.n_crops, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    TTUP_HIP_CHECK(hipMemcpyAsync(info_dev + 1, net->cert.stats + 6, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));      // low word (little endian)
    TTUP_HIP_CHECK(hipEventRecord(sl.read_info, (hipStream_t)stream));
    return TTUP_OK;
}

namespace ttup { namespace {
__global__ void cert_status_copy_kernel(const int* __restrict__ src, int* __restrict__ dst, int n, int mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] & mask;
}
int copy_status(ttup_wasb* net, int batch, int* status_dev, int mask, hipStream_t st, const char* who) {
    TTUP_REQUIRE(net && status_dev, TTUP_EINVAL, "%s: null pointer", who);
    TTUP_REQUIRE(net->cert.enabled, TTUP_EINVAL, "%s: the certified argmax is not enabled on this handle", who);
    TTUP_REQUIRE(batch >= 0 && batch <= net->max_batch * net->n_out, TTUP_EINVAL, "%s: %d heatmaps outside [0,%d]", who, batch, net->max_batch * net->n_out);
    CertState::Slot& sl = net->cert.slot[net->cert.cur];
    if (batch > 0) {
        hipLaunchKernelGGL(cert_status_copy_kernel, dim3(cdiv(batch, 256)), dim3(256), 0, st, (const int*)sl.status, status_dev, batch, mask);
        TTUP_LAUNCH_CHECK();
    }
    TTUP_HIP_CHECK(hipEventRecord(sl.read_status, st));
    return TTUP_OK;
}
} }

// 0 / 1 / 2 per heatmap, as in ABI version 100 (the guard bit is NOT part of this value: callers compare it with 1 and 2)
extern "C" int ttup_wasb_certify_status(ttup_wasb* net, int batch, int* status_dev, void* stream) {
    return copy_status(net, batch, status_dev, 3, (hipStream_t)stream, "ttup_wasb_certify_status");
}
// fp32 top-2 margin among the candidates of every heatmap of the last forward (+inf for single-candidate / unresolved heatmaps)
extern "C" int ttup_wasb_certify_margins(ttup_wasb* net, int batch, float* margin_dev, void* stream) {
    TTUP_REQUIRE(net && margin_dev, TTUP_EINVAL, "ttup_wasb_certify_margins: null pointer");
    TTUP_REQUIRE(net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_margins: the certified argmax is not enabled on this handle");
    TTUP_REQUIRE(batch >= 0 && batch <= net->max_batch * net->n_out, TTUP_EINVAL, "ttup_wasb_certify_margins: %d heatmaps outside [0,%d]", batch, net->max_batch * net->n_out);
    CertState::Slot& sl = net->cert.slot[net->cert.cur];
    if (batch > 0) TTUP_HIP_CHECK(hipMemcpyAsync(margin_dev, sl.margin, (size_t)batch * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    TTUP_HIP_CHECK(hipEventRecord(sl.read_margin, (hipStream_t)stream));
    return TTUP_OK;
}
// status | 4 where the guard band is not empty
extern "C" int ttup_wasb_certify_flags(ttup_wasb* net, int batch, int* flags_dev, void* stream) {
    return copy_status(net, batch, flags_dev, 7, (hipStream_t)stream, "ttup_wasb_certify_flags");
}

extern "C" int ttup_wasb_certify_stats(ttup_wasb* net, long long* out_host, int reset) {
    TTUP_REQUIRE(net && out_host, TTUP_EINVAL, "ttup_wasb_certify_stats: null pointer");
    TTUP_REQUIRE(net->cert.enabled, TTUP_EINVAL, "ttup_wasb_certify_stats: the certified argmax is not enabled on this handle");
    TTUP_HIP_CHECK(hipDeviceSynchronize());
    TTUP_HIP_CHECK(hipMemcpy(out_host, net->cert.stats, 12 * sizeof(long long), hipMemcpyDeviceToHost));
    if (reset) TTUP_HIP_CHECK(hipMemset(net->cert.stats, 0, 12 * sizeof(long long)));
    return TTUP_OK;
}
