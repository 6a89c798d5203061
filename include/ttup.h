/*
 * ttup.h -- C ABI of libttup.so: the MI355X (gfx950) implementation of the
 * detect -> refine -> uplift hot path of KieDani/UpliftingTableTennis.
 *
 * The reference has no native code on this path: its boundary is three Python call sites
 * (SURVEY.md 8b).  Each entry point below names the reference call it replaces; the ctypes
 * binding a maintainer would add on the reference side is shown in INTEGRATION.md and is what
 * upliftingtabletennis_amd/_lib.py contains.
 *
 * Conventions
 *   - every pointer marked "dev" is device memory on the current HIP device, owned by the caller
 *     (torch tensors: tensor.data_ptr()), alive until `stream` has been synchronised;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream), 0 = default;
 *   - all functions return 0 on success, a TTUP_E* code otherwise; ttup_last_error() returns the
 *     thread-local message of the last failure;
 *   - handles own only their packed weights and scratch; create/destroy synchronise, forward never does;
 *   - no entry point falls back to the CPU: without a usable HIP device every compute call fails.
 */
#ifndef TTUP_H
#define TTUP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTUP_OK          0
#define TTUP_EINVAL      1   /* bad argument (Python shim raises ValueError)            */
#define TTUP_EFORMAT     2   /* weight blob does not match the architecture             */
#define TTUP_EHIP        3   /* HIP runtime error (RuntimeError)                        */
#define TTUP_ENOMEM      4
#define TTUP_EMASK       5   /* uplift mask is not {0,1} with at least one 0: the reference's
                                ValueError at uplifting/model.py:541-546                */

/* activation / arithmetic type of the CNN */
#define TTUP_DTYPE_BF16  0   /* bf16 storage, bf16 MFMA, fp32 accumulate (production)   */
#define TTUP_DTYPE_F32   1   /* fp32 storage and arithmetic (parity / debug path)       */

/* refine variants */
#define TTUP_REFINE_BALL  0  /* balldetection/helper_balldetection.py:29-110            */
#define TTUP_REFINE_TABLE 1  /* tabledetection/helper_tabledetection.py:50-156          */

/* channel order flag of ttup_preprocess */
#define TTUP_LAYOUT_NCHW_F32 0
#define TTUP_LAYOUT_NHWC16   1

typedef struct ttup_wasb   ttup_wasb;
typedef struct ttup_uplift ttup_uplift;

int         ttup_version(void);
/* 16 hex digits: the hash of the sources (every .hip and .h under csrc/, include/ttup.h) and compiler flags this library was built
 * from, compiled in by upliftingtabletennis_amd/build.py (source_id()).  The Python binding refuses a library whose id differs
 * from its source tree's, so a run proves which kernels it ran. */
const char* ttup_build_id(void);
const char* ttup_last_error(void);
/* number of visible HIP devices, <0 on error; does not initialise a context */
int         ttup_device_count(void);

/* ---------------------------------------------------------------- a1: pre-processing
 * Replaces balldetection/transforms.py:17-52 (cv2.resize, INTER_LINEAR, uint8), :379-402
 * (x/255, ImageNet mean/std) and the triple concat of interface.py:104-112.
 * frames_dev: (n_frames, src_h, src_w, 3) uint8, channel order as delivered (BGR on the hub surface).
 * Triple t uses frames t, t+1, t+2 -> n_triples = n_frames-2 outputs.
 * out_dev: float32 (n_triples, 9, dst_h, dst_w) NCHW -- the tensor interface.py:112 feeds the model.
 */
int ttup_preprocess_triples(const uint8_t* frames_dev, int n_frames, int src_h, int src_w,
                            int dst_h, int dst_w, float* out_dev, void* stream);

/* single frames (table detector, tabledetection/transforms.py:17-38 + NormalizeImage; interface.py:160-165):
 * frames_dev (n,src_h,src_w,3) uint8 -> out_dev float32 (n,3,dst_h,dst_w) */
int ttup_preprocess_frames(const uint8_t* frames_dev, int n_frames, int src_h, int src_w,
                           int dst_h, int dst_w, float* out_dev, void* stream);

/* ---------------------------------------------------------------- a2: WASB / HRNet CNN
 * Replaces WASBNet.forward (balldetection/models/wasb.py:596-608) behind `self.model(x)`
 * (interface.py:115, inference/utils.py:57).
 *
 * blob: weights serialised by upliftingtabletennis_amd.weights.pack_wasb_blob:
 *   char magic[8]="TTUPWSB1"; int32 n_convs, in_ch, head_out, 0;
 *   per conv (order = reference state_dict order): int32 cout,cin,k,stride,has_bn,has_bias,0,0;
 *   float w[cout][cin][k][k]; if has_bias float b[cout]; if has_bn float gamma,beta,mean,var [cout] each.
 * BN folding (eps 1e-5) and MFMA fragment packing happen inside create.
 * height/width: network input size (multiples of 8); max_batch: largest B accepted by forward.
 * The same entry points serve the table-keypoint HRNet (tabledetection/models/hrnet.py:510-589): a blob with
 * in_ch=3, head_out=13 gives a handle whose forward takes (B,3,H,W) and returns all 13 heatmap channels (B,13,H,W);
 * argmax/win outputs then hold B*13 entries.  head_out=3 (ball) returns only channel 1, as WASBNet.forward does.
 */
int  ttup_wasb_create(const void* blob, size_t blob_bytes, int height, int width, int max_batch,
                      int dtype, ttup_wasb** out);
/* The same with the batching fixed by the caller: forward splits a batch into micro-batches of `micro_batch` samples and runs them
 * round-robin on `lanes` internal streams (0 = the defaults: 8 and 2).  lanes = 1: everything on the caller's stream -- for a
 * handle that runs side by side with another busy handle (the hub pipeline's table and ball detectors). */
int  ttup_wasb_create_ex(const void* blob, size_t blob_bytes, int height, int width, int max_batch,
                         int dtype, int micro_batch, int lanes, ttup_wasb** out);
void ttup_wasb_destroy(ttup_wasb* net);
/* The handle's internal HIP streams -> out[0..*n_out): the lane streams (none for a single-lane handle), then the stream of the
 * certified argmax's fp32 passes when it is enabled.  HIP maps all streams of a process onto a few hardware queues; work on two
 * streams of one queue never overlaps (tools/queue_probe.py measures which ones do). */
int  ttup_wasb_streams(ttup_wasb* net, void** out, int cap, int* n_out);

/* x_dev: float32 (B,9,H,W) NCHW.  heat_dev: float32 (B,1,H,W) (nullable).
 * argmax_dev: int64 (B) flat index of the first maximum of each heatmap (nullable).
 * win_dev: float32 (B,9) zero-padded 3x3 window around it (nullable). */
int ttup_wasb_forward(ttup_wasb* net, const float* x_dev, int batch, float* heat_dev,
                      int64_t* argmax_dev, float* win_dev, void* stream);

/* Fused fast path: uint8 frames in, heatmaps / peaks out (pre-processing runs inside).
 * frames_dev: (n_frames, src_h, src_w, 3) uint8; produces n_frames-2 heatmaps (triples t,t+1,t+2). */
int ttup_wasb_forward_frames(ttup_wasb* net, const uint8_t* frames_dev, int n_frames, int src_h, int src_w,
                             float* heat_dev, int64_t* argmax_dev, float* win_dev, void* stream);

/* debug/test: copy an internal activation ("stem1","stem2","layer1","trans1_0","trans1_1","stage2_0",...,
 * "stage4_0") of the last forward as float32 NCHW into out_dev; *c,*h,*w receive its shape. */
int ttup_wasb_read_tap(ttup_wasb* net, const char* name, int batch, float* out_dev, int* c, int* h, int* w, void* stream);

/* measurement aids (bench.py), HIP events on `stream`; see csrc/wasb_net.hip.  time_ops: every op on its own (warm inputs);
 * time_graph: the whole graph in launch order with an event between consecutive ops (the cache state the forward pass sees),
 * names_out = max_ops x 64 chars receiving the HIP kernel name of each op (nullable). */
int ttup_wasb_time_ops(ttup_wasb* net, int batch, int reps, int max_ops, float* ms_out, int* info_out, int* n_ops_out, void* stream);
int ttup_wasb_time_graph(ttup_wasb* net, int batch, int reps, int max_ops, float* ms_out, int* info_out, char* names_out,
                         int* n_ops_out, void* stream);
/* time_replay: the same launches back to back, `reps` passes between one pair of events (none between the ops); ms_out[0] = average
 * duration of a pass of the graph over one micro-batch (ABI 103). */
int ttup_wasb_time_replay(ttup_wasb* net, int batch, int reps, float* ms_out, void* stream);
/* measured peaks of the device (csrc/peaks.hip; SURVEY 8d "Peaks to divide by"), HIP events on `stream`, synchronises.
 * mfma: out_host[0..1] = TFLOP/s of register-resident v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 loops on random
 * operands with `waves_per_simd` waves on every SIMD, [2..3] their durations (ms).
 * hbm: out_host[0..2] = GB/s of a streaming read / copy / triad over arrays of `bytes` each (>= 1 GiB: past the Infinity Cache),
 * [3..5] their durations (ms). */
int ttup_peak_mfma_bf16(int iters, int waves_per_simd, double* out_host, void* stream);
int ttup_peak_hbm(size_t bytes, double* out_host, void* stream);
/* Certified argmax for the bf16 ball detector (north_star: bit-exact heatmap argmax indices; reference
 * balldetection/helper_balldetection.py:50 takes torch.argmax of the fp32 heatmap).  eps_abs bounds |bf16 heatmap - fp32 heatmap|
 * (calibrated by the caller on its own frames; upliftingtabletennis_amd.wasb.WASBNet.calibrate).  Once set, every forward that
 * returns peaks re-evaluates the pixels within 2*eps_abs of the bf16 maximum on fp32 receptive-field crops (crop x crop pixels,
 * 0 = 168 = the 72-pixel receptive-field radius on both sides of a 24-pixel core; a multiple of 8, at least 160 (ABI 103); at most max_crops_per_map per heatmap, 0 = 8) inside the same call, without host synchronisation, and returns the
 * fp32 winner and its fp32 3x3 window.  eps_abs < 0 switches it off.  csrc/certify.hip.
 * status (after a forward, per heatmap; ttup_wasb_certify_status): 0 = one candidate (the bf16 index is certain), 1 = resolved on
 * fp32 crops, 2 = not certified (candidate / crop budget exceeded; the bf16 index is returned).
 * flags (ttup_wasb_certify_flags): the status with bit 2 (value 4) set when the GUARD band -- the pixels between 2*eps_abs and
 * 2*1.25*eps_abs below the maximum -- is not empty: a heatmap WITHOUT that bit has the same candidates, and so the same certified
 * result, under any eps up to 1.25*eps_abs (a caller that widens eps re-runs only the heatmaps with it).
 * Both copies (and ttup_wasb_certify_info) belong to the handle's LAST forward and are ordered by the library against the next
 * forward that reuses the per-call slot, whatever stream that one is issued on.
 * stats (cumulated, synchronises): {heatmaps, single-candidate, resolved, not certified, crops, candidates of resolved,
 * bits of max |bf16 - fp32| seen at a candidate (a float in the low 32 bits: the free part of the eps audit), single-candidate
 * heatmaps cropped in exact-window mode}. */
int ttup_wasb_set_certify(ttup_wasb* net, float eps_abs, int crop, int max_crops_per_map);
/* step 1 of the certification on its own (the kernel that streams the heatmap on the production path: measurement aid, test
 * hook): for each of n_maps fp32 heatmaps (n_maps, H, W) the pixels whose value is >= heat[argmax[map]] - 2*eps_abs, appended in
 * arbitrary order to cand_idx_dev[map*K ..] (flat pixel index) / cand_bf_dev (their values); cand_cnt_dev[map] (zeroed by the
 * caller) counts ALL of them, also those past K. */
int ttup_certify_scan(const float* heat_dev, const int64_t* argmax_dev, int n_maps, int height, int width, float eps_abs, int K,
                      int* cand_idx_dev, int* cand_cnt_dev, float* cand_bf_dev, void* stream);
/* max |a - b| over n floats -> out_dev[0] (device, written on `stream`; NaN anywhere gives NaN).  The eps audit's error measure
 * (upliftingtabletennis_amd/wasb.py: heatmap_error): one pass over the two heatmaps instead of three torch kernels, and free of
 * packed fp32 instructions, which must not run beside the CNN (csrc/common.h). */
int ttup_max_abs_diff(const float* a_dev, const float* b_dev, long long n, float* out_dev, void* stream);
/* the same over the columns [c0, c1) of `rows` rows of `width` floats (a strip audit leaves out the columns whose receptive field
 * reaches the strip's artificial border); accumulate != 0 keeps the running maximum already in out_dev[0] */
int ttup_max_abs_diff_cols(const float* a_dev, const float* b_dev, long long rows, long long width, long long c0, long long c1,
                           float* out_dev, int accumulate, void* stream);
/* dst[r][j] = src[r][x0 + j] for r < rows, j < w: a column strip of a (rows, width) float array (the audit's strip of the
 * pre-processed input, without torch's copy kernels on the audit stream) */
int ttup_slice_columns(const float* src_dev, long long rows, int width, int x0, int w, float* dst_dev, void* stream);
/* exact-window mode (on != 0): heatmaps with ONE candidate get an fp32 crop too, so that every returned 3x3 window -- not only
 * the near-ties' -- holds the fp32 path's values and the sub-pixel fit sees what the reference's fit sees (one 168x168 fp32
 * pass per heatmap: a parity / audit mode, off by default) */
int ttup_wasb_certify_exact_windows(ttup_wasb* net, int on);
/* audit crops (every > 0): of the frames f of the following forwards with (f + phase) % every == 0, ONE single-candidate heatmap
 * (channel (f / every) % channels) gets an fp32 crop although its index is already certain; like every crop it reports
 * |bf16 - fp32| at its candidate into the running maximum that ttup_wasb_certify_info returns.  This is the candidate-level audit
 * extended to heatmaps WITHOUT near-ties (reference: helper_balldetection.py:50 takes the argmax of an fp32 heatmap; the bound on
 * |bf16 - fp32| is what lets the bf16 path return the same index).  every = 0 switches it off (default). */
int ttup_wasb_certify_audit_crops(ttup_wasb* net, int every, int phase);
/* info_dev[0] = crops the LAST forward asked for (may exceed the budget it had); info_dev[1] = the bits of a float: the largest
 * |bf16 - fp32| seen so far at any candidate of any call (stats[6]); both copied in stream order, no synchronisation */
int ttup_wasb_certify_info(ttup_wasb* net, int* info_dev, void* stream);
int ttup_wasb_certify_status(ttup_wasb* net, int batch, int* status_dev, void* stream);
int ttup_wasb_certify_flags(ttup_wasb* net, int batch, int* flags_dev, void* stream);
/* measurement: by how much the fp32 winner of each heatmap of the last forward leads the best OTHER candidate on the fp32 crops
 * (float32 (batch); +inf where there was one candidate or the heatmap was not resolved).  A margin below the accuracy of the fp32
 * path against the reference's own fp32 arithmetic marks a heatmap whose argmax the reference itself does not determine. */
int ttup_wasb_certify_margins(ttup_wasb* net, int batch, float* margin_dev, void* stream);
/* crops the following forward calls may use (default: max_batch, i.e. one per heatmap): the call enqueues ceil(budget / 64) fp32
 * passes sized on the device, so a caller that knows its typical crop count (stats / status of earlier calls) saves the empty
 * passes; heatmaps beyond the budget are flagged 2 (64: 128 since ABI 103, TTUP_CERT_CH) */
int ttup_wasb_certify_budget(ttup_wasb* net, int max_crops);
/* running counters of the handle since creation / the last reset, copied to TWELVE long longs on the host (synchronises): [0] heatmaps,
 * [1] single candidate, [2] resolved on fp32 crops, [3] not certified (= [8] + [9] + [10]), [4] crops, [5] candidates of the resolved
 * heatmaps, [6] low word = bits of the largest |bf16 - fp32| seen at a candidate, [7] single-candidate heatmaps that got a crop
 * (exact-window mode / audit crops), [8] not certified: candidate list overflow, [9]: more crops than one heatmap / frame may add,
 * [10]: the call's crop list was full, [11] crops of class 2 among [4] (candidates within a 14-position core: pruned to a smaller cone;
 * ABI 103).  (ABI version 100 copied eight.) */
int ttup_wasb_certify_stats(ttup_wasb* net, long long* out_host12, int reset);
/* scheduling priority of the handle's internal streams (high != 0: greatest device priority); synchronises */
int ttup_wasb_set_priority(ttup_wasb* net, int high);
/* micro-batch the handle was created with (TTUP_MICRO_BATCH) */
int ttup_wasb_micro_batch(ttup_wasb* net);
/* heatmap channels per sample returned by forward: 1 (ball) or 13 (table) */
int ttup_wasb_out_channels(ttup_wasb* net);

/* ---------------------------------------------------------------- a3/a4: heatmap argmax + refine
 * Replaces extract_position_torch_gaussian (ball: helper_balldetection.py:29-110, called at
 * inference/utils.py:59; table: helper_tabledetection.py:50-156, called at interface.py:116).
 * heat_dev: float32 (n_maps, H, W).  out_xyv_dev: float64 (n_maps,3) [x,y,visibility] in
 * img_w x img_h pixel coordinates.  argmax_dev / win_dev as above (nullable outputs).
 * ws_dev: scratch of at least ttup_refine_workspace_bytes(n_maps,H,W) bytes.
 */
size_t ttup_refine_workspace_bytes(int n_maps, int height, int width);
int ttup_refine(const float* heat_dev, int n_maps, int height, int width, int img_w, int img_h, int variant,
                double* out_xyv_dev, int64_t* argmax_dev, float* win_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* second half only: peaks already known (from ttup_wasb_forward) */
int ttup_refine_windows(const int64_t* argmax_dev, const float* win_dev, int n_maps, int height, int width,
                        int img_w, int img_h, int variant, double* out_xyv_dev, void* stream);

/* ---------------------------------------------------------------- a6: uplift transformer
 * Replaces MultiStageModel.forward (uplifting/model.py:529-571, 'connectstage', mode 'dynamic',
 * time_rotation 'new') behind `self.model(ball, table, mask, times)` (interface.py:235, inference/utils.py:254).
 * blob: upliftingtabletennis_amd.weights.pack_uplift_blob:
 *   char magic[8]="TTUPUPL1"; int32 dim, heads, n_pos_layers, n_first_layers, n_second_layers, n_table, 0, 0;
 *   then every tensor of the state_dict (reference order, without embed.* and inv_freq): int32 numel; float data[numel].
 */
int  ttup_uplift_create(const void* blob, size_t blob_bytes, int max_batch, int max_len, ttup_uplift** out);
void ttup_uplift_destroy(ttup_uplift* net);
/* ball (B,T,2), table (B,13,3), mask (B,T) in {0,1}, times (B,T) seconds -> rot (B,3), pos (B,T,3); all float32 dev.
 * check_mask != 0 reproduces the reference's ValueError (TTUP_EMASK) and costs one stream synchronisation. */
int ttup_uplift_forward(ttup_uplift* net, const float* ball_dev, const float* table_dev, const float* mask_dev,
                        const float* times_dev, int batch, int len, float* rot_dev, float* pos_dev,
                        int check_mask, void* stream);

/* Small batches (batch * len <= 1024 tokens) are replayed from a hipGraph captured on the second call with a given (batch, len).
 * out_host3[0] = graphs held, [1] = 1 when the graph path is off (capture failed on this runtime, or TTUP_UPLIFT_NO_GRAPH is set),
 * [2] = forwards served by a replay so far. */
int ttup_uplift_graph_info(ttup_uplift* net, int* out_host3);

/* Sequences of at most 64 tokens (the table stage always; the temporal and spin stages of clips of up to 63 frames) run ALL layers
 * of a stage in one launch with the tokens resident on the CU (TTUP_UPLIFT_NO_STAGE=1 switches it off, TTUP_UPLIFT_STAGE_WG caps the
 * launch size it is used for).  *out_host = such launches issued or captured so far. */
int ttup_uplift_stage_info(ttup_uplift* net, long long* out_host);

/* ---------------------------------------------------------------- a7: spin frame change
 * Replaces transform_rotationaxes (uplifting/helper.py:394-420): rot (B,3), pos (B,T,3) -> out (B,3). */
int ttup_transform_rotationaxes(const float* rot_dev, const float* pos_dev, int batch, int len, float* out_dev, void* stream);

/* ---------------------------------------------------------------- f2: synthetic-trajectory generator (BASELINE config 5)
 * Replaces the per-seed work of find_valid_trajectories_worker (syntheticdataset/mujocosimulation.py:112-219):
 *   ttup_trajgen_simulate  = _init_simulation (:54-109, CPython random.Random(seed) reproduced bit for bit) + the
 *                            mj_step sampling loop with its out-of-bounds / out-of-image stops (:112-151);
 *   ttup_trajgen_select    = _count_hits (helper.py:282-321) + every rejection and cut rule (:152-219).
 * mode: 0 final_lose, 1 final_win, 2 intermediate, 3 first_good, 4 first_short, 5 first_long (OOB_DEFINITIONS order);
 * direction: 0 left_to_right, 1 right_to_left.  substeps = RK4 steps per 1 ms MuJoCo timestep.
 * cam_host: 25 doubles on the HOST, Mext (4x4 row major) then Mint (3x3), as _calc_cammatrices builds them.
 * samples_dev: [n_labels][9][n_seeds] doubles (x y z vx vy vz wx wy wz; n_labels = ttup_trajgen_max_samples());
 * n_saved_dev: valid samples per seed.  init_dev (optional): [9][n_seeds] initial states.
 * n_keep_dev: 0 = rejected, else the number of samples the reference keeps; bounces_dev [n_seeds][4], n_bounces_dev.
 * The fluid / contact arithmetic restates MuJoCo's documented model (MuJoCo itself is not available): parity unpinned. */
int    ttup_trajgen_max_samples(void);
size_t ttup_trajgen_workspace_bytes(int n_seeds);
int ttup_trajgen_simulate(const int64_t* seeds_dev, int n_seeds, int mode, int direction, int substeps, const double* cam_host,
                          double* samples_dev, int* n_saved_dev, double* init_dev, void* workspace, size_t workspace_bytes, void* stream);
int ttup_trajgen_select(const double* samples_dev, const int* n_saved_dev, int n_seeds, int mode, int direction,
                        int* n_keep_dev, double* bounces_dev, int* n_bounces_dev, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- f4: camera calibration (csrc/calib.hip)
 * Replaces calibrate_camera (inference/utils.py:312-329 -> dataprocessing/regress_cameramatrices.py:199-231 with
 * use_ransac=True, dataprocessing/my_dlt.py) behind TableDetector.calibrate_camera / TableTennisPipeline.calibrate_camera
 * (interface.py:174-175, :291-299).  One workgroup per camera: DLT start, one lane per RANSAC subset, refinement on the inliers.
 * keypoints (B,13,3) float64 [x, y, visibility]; subsets (B,n_subsets,4) int32: the keys (1..13) drawn per RANSAC iteration
 * (keypoints 10 and 11 are added to every subset on the device); img_w/img_h: the principal point is fixed at (w//2, h//2).
 * Outputs: mint (B,3,4), mext (B,4,4) float64 row-major, n_inliers (B), status (B): 0 ok, -1 fewer than 6 visible keypoints,
 * -2 degenerate DLT start, -3 no inliers; start (B,8) the DLT start (fx, fy, t, euler xyz), nullable (tests). */
int ttup_calib_forward(const double* keypoints_dev, const int* subsets_dev, int batch, int n_subsets, int img_w, int img_h, int max_iter,
                       double* mint_dev, double* mext_dev, int* n_inliers_dev, int* status_dev, double* start_dev, void* stream);

/* ---------------------------------------------------------------- g1: drag + Magnus ODE fit (extension; csrc/odefit.hip)
 * BASELINE.json north_star names a "batched RK4 + Jacobian/Gauss-Newton" fit of flight dynamics to the detected 2-D track.
 * The reference has NO such code (its uplift is the transformer above): these entry points replace nothing and
 * are never called by the drop-in surface; parity is unpinned, validation is by self-consistency.
 * All arrays float64 on the device.  obs_xy (B,T,2) pixels; times (B,T) seconds, non-decreasing; mask (B,T) 0/1 or null;
 * cam: (B or 1, 21) = rows 0..2 of Mext (world -> camera, 12 numbers) then Mint (9), one per trajectory when cam_per_traj != 0;
 * init (B,9) starting point (r0 [m], v0 [m/s], w0 [rad/s] at times[:,0]).  Outputs: params (B,9), pos3d (B,T,3) at the time
 * stamps, cost (B) mean squared reprojection error [px^2], iters (B) accepted Levenberg-Marquardt steps (nullable except params).
 * h_max: largest RK4 step; an interval between two time stamps is cut into ceil(dt / h_max) equal steps. */
int ttup_odefit_forward(const double* obs_xy_dev, const double* times_dev, const double* mask_dev, const double* cam_dev, int cam_per_traj,
                        const double* init_dev, int batch, int len, double h_max, int max_iter, double tol,
                        double* params_dev, double* pos3d_dev, double* cost_dev, int* iters_dev, void* stream);
/* forward model only: params (B,9) -> pos3d (B,T,3) and / or pixels px (B,T,2) (either may be null) */
int ttup_odefit_integrate(const double* params_dev, const double* times_dev, const double* cam_dev, int cam_per_traj, int batch, int len,
                          double h_max, double* pos3d_dev, double* px_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTUP_H */
