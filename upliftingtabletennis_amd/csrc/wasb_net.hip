// a2: the WASB / HRNet ball-heatmap CNN as a static op list over NHWC buffers.
// Graph follows balldetection/models/wasb.py: HRNet.forward :445-486, HighResolutionModule.forward :227-245,
// fuse construction :179-222, transitions :362-396, config :514-573, WASBNet.forward :596-608.
//
// What is different from the reference's eager module tree (results unchanged):
//   * BatchNorm (eval) is folded into every conv at create time;
//   * Bottleneck conv3 (1x1 32->128) and its 1x1 downsample (64->128) + add + ReLU run as ONE two-source
//     1x1 conv with K = 32+64 (the 128-channel pre-activation never touches HBM);
//   * stage-4 fused outputs 1..3 (only consumed when classify_invisible=True, never set by get_model,
//     balldetection/train.py:268) and head channels 0 and 2 (dropped at wasb.py:606) are not computed;
//   * the batch is processed in micro-batches so that intermediate tensors stay near the Infinity Cache.
#include "wasb_net.h"
#include <string.h>
#include <stdlib.h>
#include <memory>

using namespace ttup;

namespace {

const int STAGE_CH[4] = {16, 32, 64, 128};

struct BlobReader {
    const char* p; size_t left;
    bool read(void* dst, size_t n) { if (n > left) return false; memcpy(dst, p, n); p += n; left -= n; return true; }
};

}  // namespace

ttup_wasb::~ttup_wasb() {
    cert_free(this);
    for (auto& c : convs) free_conv(&c);
    if (lanes.empty()) {       // construction failed before the lanes were set up
        for (auto& t : tensors) if (t.ptr) (void)hipFree(t.ptr);
    }
    for (auto& L : lanes) {
        for (void* q : L.ptr) if (q) (void)hipFree(q);
        if (L.heat_scratch) (void)hipFree(L.heat_scratch);
        if (L.refine_ws) (void)hipFree(L.refine_ws);
        if (L.argmax_scratch) (void)hipFree(L.argmax_scratch);
        if (L.win_scratch) (void)hipFree(L.win_scratch);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.done) (void)hipEventDestroy(L.done);
    }
    if (fork) (void)hipEventDestroy(fork);
    if (pass_done) (void)hipEventDestroy(pass_done);
    if (head_w_dev) (void)hipFree(head_w_dev);
    if (head_b_dev) (void)hipFree(head_b_dev);
}

namespace {

// ---- parse the blob into folded convs (order = upliftingtabletennis_amd.arch.hrnet_convs)
int parse_blob(const void* blob, size_t bytes, std::vector<FoldedConv>* out, int* in_ch, int* head_out,
               std::vector<float>* head_w, std::vector<float>* head_b) {
    BlobReader r{(const char*)blob, bytes};
    char magic[8]; int hdr[4];
    TTUP_REQUIRE(r.read(magic, 8) && memcmp(magic, "TTUPWSB1", 8) == 0, TTUP_EFORMAT, "wasb blob: bad magic");
    TTUP_REQUIRE(r.read(hdr, sizeof hdr), TTUP_EFORMAT, "wasb blob: truncated header");
    const int n = hdr[0];
    *in_ch = hdr[1]; *head_out = hdr[2];
    TTUP_REQUIRE(n == 72, TTUP_EFORMAT, "wasb blob: expected 72 convs, got %d", n);
    for (int i = 0; i < n; ++i) {
        int h[8];
        TTUP_REQUIRE(r.read(h, sizeof h), TTUP_EFORMAT, "wasb blob: truncated at conv %d", i);
        FoldedConv c; c.cout = h[0]; c.cin = h[1]; c.k = h[2]; c.stride = h[3];
        const int has_bn = h[4], has_bias = h[5];
        TTUP_REQUIRE(c.cout > 0 && c.cout <= 128 && c.cin > 0 && c.cin <= 128 && (c.k == 1 || c.k == 3), TTUP_EFORMAT,
                     "wasb blob: conv %d has unsupported shape %dx%dx%d", i, c.cout, c.cin, c.k);
        const size_t nw = (size_t)c.cout * c.cin * c.k * c.k;
        c.w.resize(nw); c.bias.assign(c.cout, 0.f);
        TTUP_REQUIRE(r.read(c.w.data(), nw * 4), TTUP_EFORMAT, "wasb blob: truncated weights of conv %d", i);
        if (has_bias) TTUP_REQUIRE(r.read(c.bias.data(), c.cout * 4), TTUP_EFORMAT, "wasb blob: truncated bias of conv %d", i);
        if (has_bn) {
            std::vector<float> bn(4 * c.cout);
            TTUP_REQUIRE(r.read(bn.data(), bn.size() * 4), TTUP_EFORMAT, "wasb blob: truncated BN of conv %d", i);
            const float *gamma = bn.data(), *beta = gamma + c.cout, *mean = beta + c.cout, *var = mean + c.cout;
            const size_t per = (size_t)c.cin * c.k * c.k;
            for (int o = 0; o < c.cout; ++o) {
                // y = (conv(x)+b - mean) * gamma / sqrt(var + eps) + beta, eps = 1e-5 (nn.BatchNorm2d default)
                const double s = (double)gamma[o] / sqrt((double)var[o] + 1e-5);
                for (size_t j = 0; j < per; ++j) c.w[o * per + j] = (float)((double)c.w[o * per + j] * s);
                c.bias[o] = (float)(((double)c.bias[o] - (double)mean[o]) * s + (double)beta[o]);
            }
        }
        out->push_back(std::move(c));
    }
    TTUP_REQUIRE(r.left == 0, TTUP_EFORMAT, "wasb blob: %zu trailing bytes", r.left);
    const FoldedConv& head = out->back();
    TTUP_REQUIRE(head.k == 1 && head.cin == 16 && head.cout == *head_out, TTUP_EFORMAT, "wasb blob: unexpected head shape");
    *head_w = head.w; *head_b = head.bias;
    return TTUP_OK;
}

struct Builder {
    ttup_wasb* net;
    const std::vector<FoldedConv>* folded;
    size_t cursor = 0;     // next folded conv in reference order
    int rc = TTUP_OK;

    int new_tensor(int c, int h, int w) {
        Tensor t; t.c = c; t.h = h; t.w = w;
        const size_t bytes = (size_t)net->micro * h * w * c * net->esize();
        if (hipMalloc(&t.ptr, bytes) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", bytes); rc = TTUP_ENOMEM; t.ptr = nullptr; }
        net->tensors.push_back(t);
        return (int)net->tensors.size() - 1;
    }
    const FoldedConv& next(int cout, int cin, int k, int stride) {
        const FoldedConv& f = (*folded)[cursor++];
        if (f.cout != cout || f.cin != cin || f.k != k || f.stride != stride) {
            set_error("wasb blob: conv %zu is %dx%dx%d/s%d, architecture expects %dx%dx%d/s%d", cursor - 1, f.cout, f.cin, f.k, f.stride, cout, cin, k, stride);
            rc = TTUP_EFORMAT;
        }
        return f;
    }
    int pack(const FoldedConv& a, const FoldedConv* b, int cin_pad) {
        PackedConv p;
        if (rc == TTUP_OK) { const int e = pack_conv(a, b, cin_pad, net->dtype, &p); if (e) rc = e; }
        net->convs.push_back(p);
        return (int)net->convs.size() - 1;
    }
    // conv op on tensor `src` -> new tensor
    int conv(int src, int cout, int k, int stride, int relu, int residual = -1, int dst = -1) {
        const Tensor s = net->tensors[src];
        const FoldedConv& f = next(cout, s.c, k, stride);
        const int pc = pack(f, nullptr, s.c);
        if (dst < 0) dst = new_tensor(cout, (s.h + stride - 1) / stride, (s.w + stride - 1) / stride);
        Op op; op.kind = Op::CONV; op.conv = pc; op.src0 = src; op.residual = residual; op.dst = dst; op.relu = relu;
        net->ops.push_back(op);
        return dst;
    }
    int basic_block(int x) {       // wasb.py:48-64
        const int c = net->tensors[x].c;
        const int t = conv(x, c, 3, 1, 1);
        return conv(t, c, 3, 1, 1, /*residual*/ x);
    }
    Op bb_chain_op(int x, int n_convs, bool need_dst) {      // n_convs/2 BasicBlocks fused (bf16 path); not yet in the op list
        const Tensor s = net->tensors[x];
        Op op; op.kind = Op::BB_CHAIN; op.src0 = x; op.n_chain = n_convs;
        for (int i = 0; i < n_convs; ++i) op.chain[i] = pack(next(s.c, s.c, 3, 1), nullptr, s.c);
        op.dst = need_dst ? new_tensor(s.c, s.h, s.w) : -1;
        return op;
    }
    int bb_chain(int x, int n_convs) {
        const Op op = bb_chain_op(x, n_convs, true);
        net->ops.push_back(op);
        return op.dst;
    }
    // HighResolutionModule (wasb.py:227-245); returns fused outputs 0..n_out-1
    std::vector<int> stage(std::vector<int> xs, int n_out, bool head_mode = false) {
        const int nb = (int)xs.size();
        const bool fuse = net->dtype == TTUP_DTYPE_BF16 && !getenv("TTUP_NO_FUSE");
        // the full-resolution branch's fuse-layer sum rides in the epilogue of its two-block chain, which is therefore emitted
        // AFTER the lower branches and their 1x1 fuse convs (deferred below); its weights are still consumed in reference order
        const bool fuse_sum = fuse && !getenv("TTUP_NO_FUSE_SUM") && net->tensors[xs[0]].c == 16;
        Op deferred; bool has_deferred = false;
        const int x0_in = xs[0];
        for (int b = 0; b < nb; ++b) {
            const Tensor xt = net->tensors[xs[b]];
            if (b == 0 && fuse_sum) { deferred = bb_chain_op(xs[0], 4, /*pre-fuse tensor has consumers*/ n_out > 1); has_deferred = true; xs[0] = deferred.dst; }
            else if (fuse && xt.c == 16) xs[b] = bb_chain(xs[b], 4);                       // both blocks in one kernel
            else if (fuse && xt.c == 32) { xs[b] = bb_chain(xs[b], 2); xs[b] = bb_chain(xs[b], 2); }
            else { xs[b] = basic_block(xs[b]); xs[b] = basic_block(xs[b]); }
        }
        // reference order of the fuse convs in the state_dict: i major, j minor, chain index k
        struct Term { int i, j; std::vector<const FoldedConv*> chain; };
        std::vector<Term> terms;
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < nb; ++j) {
                if (j == i) continue;
                Term t; t.i = i; t.j = j;
                if (j > i) t.chain.push_back(&next(STAGE_CH[i], STAGE_CH[j], 1, 1));
                else for (int k = 0; k < i - j; ++k) t.chain.push_back(&next(k == i - j - 1 ? STAGE_CH[i] : STAGE_CH[j], STAGE_CH[j], 3, 2));
                terms.push_back(t);
            }
        std::vector<int> outs;
        for (int i = 0; i < n_out; ++i) {
            const Tensor xi = net->tensors[i == 0 ? x0_in : xs[i]];       // same shape as x_i (xs[0] is -1 when the branch tensor is not stored)
            // running sum: starts at x_i (identity term) or at the j=0 chain for i>0, in reference order
            int acc = -1;
            bool acc_is_xi = false;
            std::vector<int> up_t; std::vector<int> up_s;
            int last_chain_op = -1;               // index in net->ops of the conv that completes the running sum (j = i-1 chain)
            // emission order: the 1x1 convs of the lower branches (j > i) first, so that a later conv's epilogue can add them
            for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < nb; ++j) {
                if ((pass == 0) != (j > i)) continue;
                if (j == i) {
                    if (acc < 0) { acc = xs[i]; acc_is_xi = true; }
                    else {
                        // x_i enters the sum after at least one chain term: fold it in as the residual of ... nothing to
                        // run, so add it through an UPSUM term with shift 0 below
                        up_t.push_back(xs[i]); up_s.push_back(0);
                    }
                    continue;
                }
                const Term* tm = nullptr;
                for (auto& t : terms) if (t.i == i && t.j == j) tm = &t;
                if (j > i) {        // 1x1 conv + BN at the low resolution, upsampled when summed
                    const int pc = pack(*tm->chain[0], nullptr, 0);
                    const Tensor sj = net->tensors[xs[j]];
                    const int dst = new_tensor(STAGE_CH[i], sj.h, sj.w);
                    // 32 -> 16 on the output of a fused 32-channel block: rides in that kernel's epilogue (one MFMA per pixel group)
                    bool attached = false;
                    if (fuse && sj.c == 32 && STAGE_CH[i] == 16) {
                        for (int k = (int)net->ops.size() - 1; k >= 0 && !attached; --k) {
                            Op& po = net->ops[k];
                            if (po.dst != xs[j]) continue;
                            if (po.kind == Op::BB_CHAIN && po.n_chain == 2 && po.conv2 < 0) { po.conv2 = pc; po.dst2 = dst; attached = true; }
                            break;
                        }
                    }
                    // 64 -> 16 / 64 -> 32 on the output of the branch's last 64 -> 64 conv: rides in that conv's epilogue
                    if (!attached && fuse && sj.c == 64 && (STAGE_CH[i] == 16 || STAGE_CH[i] == 32) && !getenv("TTUP_NO_FUSE_LIN")) {
                        for (int k = (int)net->ops.size() - 1; k >= 0 && !attached; --k) {
                            Op& po = net->ops[k];
                            if (po.dst != xs[j]) continue;
                            const PackedConv& pp = net->convs[po.conv >= 0 ? po.conv : 0];
                            if (po.kind == Op::CONV && po.conv >= 0 && po.conv2 < 0 && pp.k == 3 && pp.stride == 1 && pp.cout == 64 && pp.cin_total == 64 && pp.c0 == 64 && po.src1 < 0) {
                                if (STAGE_CH[i] == 16 && po.lin16 < 0) { po.lin16 = pc; po.lin16_dst = dst; attached = true; }
                                else if (STAGE_CH[i] == 32 && po.lin32 < 0) { po.lin32 = pc; po.lin32_dst = dst; attached = true; }
                            }
                            break;
                        }
                    }
                    Op op; op.kind = Op::CONV; op.conv = pc; op.src0 = xs[j]; op.dst = dst; op.relu = 0;
                    if (!attached) net->ops.push_back(op);
                    up_t.push_back(dst); up_s.push_back(j - i);
                } else {            // chain of stride-2 3x3 convs; the last one adds the running sum
                    int cur = xs[j];
                    for (size_t k = 0; k < tm->chain.size(); ++k) {
                        const bool last = k + 1 == tm->chain.size();
                        const FoldedConv& f = *tm->chain[k];
                        const int pc = pack(f, nullptr, 0);
                        const Tensor sc = net->tensors[cur];
                        const int dst = new_tensor(f.cout, (sc.h + 1) / 2, (sc.w + 1) / 2);
                        Op op; op.kind = Op::CONV; op.conv = pc; op.src0 = cur; op.dst = dst; op.relu = last ? 0 : 1;
                        if (last && acc >= 0) op.residual = acc;
                        // 16 -> 16 on the full-resolution branch while an earlier fuse chain took the same tensor down 16 -> 32: both
                        // convs in one pass over it (conv_s2_pair_kernel)
                        bool paired = false;
                        if (fuse && !last && f.cout == 16 && sc.c == 16 && !getenv("TTUP_NO_PAIR")) {
                            for (int q = (int)net->ops.size() - 1; q >= 0 && !paired; --q) {
                                Op& po = net->ops[q];
                                if (po.kind != Op::CONV || po.src0 != cur || po.conv < 0) continue;
                                const PackedConv& pp = net->convs[po.conv];
                                if (pp.k == 3 && pp.stride == 2 && pp.cout == 32 && pp.cin_total == 16 && po.pair < 0 && po.conv2 < 0 && po.src1 < 0) {
                                    po.pair = pc; po.pair_dst = dst; po.pair_relu = op.relu; paired = true;
                                }
                            }
                        }
                        if (!paired) net->ops.push_back(op);
                        if (last) last_chain_op = (int)net->ops.size() - 1;
                        cur = dst;
                    }
                    acc = cur; acc_is_xi = false;
                }
            }
            (void)acc_is_xi;
            // the reference adds the terms in branch order j (wasb.py:236-243): x_i (shift 0) before the upsampled lower branches
            for (size_t k = 1; k < up_t.size(); ++k)
                for (size_t q = k; q > 0 && up_s[q] < up_s[q - 1]; --q) { std::swap(up_s[q], up_s[q - 1]); std::swap(up_t[q], up_t[q - 1]); }
            if (i == 0 && has_deferred) {
                // y_0 = relu(x_0 + sum_j up(1x1(x_j))) in the epilogue of the branch's block chain
                deferred.n_terms = (int)up_t.size();
                if (up_t.size() > 3) { set_error("fuse: more than 3 upsample terms"); rc = TTUP_EINVAL; }
                for (size_t k = 0; k < up_t.size() && k < 3; ++k) { deferred.terms[k] = up_t[k]; deferred.shifts[k] = up_s[k]; }
                if (head_mode) { deferred.head = 1; deferred.dst2 = -1; }
                else deferred.dst2 = new_tensor(xi.c, xi.h, xi.w);
                net->ops.push_back(deferred);
                outs.push_back(deferred.dst2);
                continue;
            }
            // bf16: y_i = relu(chains + x_i + up(...)) finishes in the epilogue of the last chain conv (one same-resolution term
            // and one upsampled term fit): the element-wise pass over the branch disappears
            if (fuse && i > 0 && last_chain_op >= 0 && up_t.size() <= 2 && net->ops[last_chain_op].dst == acc) {
                int same = -1, upt = -1, ups = 0;
                bool ok = true;
                for (size_t k = 0; k < up_t.size(); ++k) {
                    if (up_s[k] == 0 && same < 0) same = up_t[k];
                    else if (up_s[k] > 0 && upt < 0) { upt = up_t[k]; ups = up_s[k]; }
                    else ok = false;
                }
                if (ok) {
                    Op& lc = net->ops[last_chain_op];
                    lc.res2 = same; lc.res3 = upt; lc.sh3 = ups; lc.relu = 1;
                    outs.push_back(lc.dst);
                    continue;
                }
            }
            // y = relu(acc + sum of upsampled / late identity terms)
            const int dst = new_tensor(xi.c, xi.h, xi.w);
            Op op; op.kind = Op::UPSUM; op.src0 = acc; op.dst = dst; op.n_terms = (int)up_t.size();
            if (up_t.size() > 3) { set_error("fuse: more than 3 upsample terms"); rc = TTUP_EINVAL; }
            for (size_t k = 0; k < up_t.size() && k < 3; ++k) { op.terms[k] = up_t[k]; op.shifts[k] = up_s[k]; }
            net->ops.push_back(op);
            outs.push_back(dst);
        }
        // convs of dead fused outputs (i >= n_out) are skipped but stay consumed from the cursor
        return outs;
    }
};

int build(ttup_wasb* net, const std::vector<FoldedConv>& folded) {
    Builder b; b.net = net; b.folded = &folded;
    const int H = net->H, W = net->W;
    net->t_input = b.new_tensor(16, H, W);
    // stem (wasb.py:446-451)
    int x;
    const bool fuse_c1 = net->dtype == TTUP_DTYPE_BF16 && !getenv("TTUP_NO_FUSE");
    int stem_a1 = -1;
    if (fuse_c1 && !getenv("TTUP_NO_STEM")) {
        // conv1 + conv2 + Bottleneck conv1 in one persistent kernel; the first 64-channel tensor never reaches HBM
        const FoldedConv& c1 = b.next(64, net->in_ch, 3, 1);
        const int p1 = b.pack(c1, nullptr, 16);
        // the same conv for the stem's frames mode: input slot f*4 + c holds colour c of frame f (slot 3 of every frame and the
        // slots past the last frame carry zero weights)
        FoldedConv c1f = c1;
        c1f.cin = 16; c1f.w.assign((size_t)64 * 16 * 9, 0.f);
        for (int co = 0; co < 64 && b.rc == TTUP_OK; ++co)
            for (int ci = 0; ci < net->in_ch; ++ci)
                for (int t = 0; t < 9; ++t) c1f.w[((size_t)co * 16 + (ci / 3) * 4 + ci % 3) * 9 + t] = c1.w[((size_t)co * net->in_ch + ci) * 9 + t];
        int p1f = b.pack(c1f, nullptr, 16);
        // ... and, for triples, the 4-k-step form of that conv (csrc/conv.hip stem_kernel<3, true>): K = 3 tap rows x 40 slots, slot
        // o of a row = pixel dx = o / 12, frame (o % 12) / 4, colour o % 4 (colour 3 and o >= 36: zero weights), packed as a
        // "1x1 conv with 128 inputs" so that k-step s, lane group g, element j holds k = 32 s + 8 g + j
        if (net->in_ch == 9 && !getenv("TTUP_STEM_K5")) {
            FoldedConv c1k = c1;
            c1k.cin = 128; c1k.k = 1; c1k.w.assign((size_t)64 * 128, 0.f);
            for (int co = 0; co < 64; ++co)
                for (int r = 0; r < 3; ++r)
                    for (int o = 0; o < 36; ++o) {
                        const int dx = o / 12, f = (o % 12) / 4, col = o % 4;
                        if (col < 3) c1k.w[(size_t)co * 128 + r * 40 + o] = c1.w[((size_t)co * net->in_ch + f * 3 + col) * 9 + r * 3 + dx];
                    }
            p1f = b.pack(c1k, nullptr, 0);
        }
        const int p2 = b.pack(b.next(64, 64, 3, 1), nullptr, 0);
        const int p3 = b.pack(b.next(32, 64, 1, 1), nullptr, 0);
        x = b.new_tensor(64, H, W); net->taps["stem2"] = x;
        stem_a1 = b.new_tensor(32, H, W);
        Op op; op.kind = Op::STEM; op.conv = p1; op.conv2 = p2; op.conv3 = p3; op.src0 = net->t_input; op.dst = x; op.dst2 = stem_a1;
        op.conv1f = p1f;
        net->ops.push_back(op);
        if (!getenv("TTUP_NO_FRAMES_MODE")) {
            // (micro + nf - 1) frames of (H, W, 4) bf16: allocated through the tensor list so that every lane gets its own copy
            Tensor t; t.c = 4; t.h = H; t.w = W; t.extra = net->in_ch / 3 - 1;
            const size_t bytes = (size_t)(net->micro + net->in_ch / 3 - 1) * H * W * 4 * 2;
            if (hipMalloc(&t.ptr, bytes) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", bytes); b.rc = TTUP_ENOMEM; t.ptr = nullptr; }
            net->tensors.push_back(t);
            net->t_frames = (int)net->tensors.size() - 1;
        }
    } else {
        {
            const FoldedConv& f = b.next(64, net->in_ch, 3, 1);
            const int pc = b.pack(f, nullptr, 16);
            const int dst = b.new_tensor(64, H, W);
            Op op; op.conv = pc; op.src0 = net->t_input; op.dst = dst; op.relu = 1; net->ops.push_back(op);
            x = dst; net->taps["stem1"] = x;
        }
        x = b.conv(x, 64, 3, 1, 1); net->taps["stem2"] = x;
    }
    const size_t stem2_op = net->ops.size() - 1;
    // layer1: Bottleneck(64 -> 32 -> 128) (wasb.py:85-105), conv3 + downsample fused into one two-source 1x1 conv;
    // transition1 (wasb.py:454-459).  bf16: both run in one kernel and the 128-channel tensor stays in LDS.
    std::vector<int> xs(2);
    {
        int a1;
        if (stem_a1 >= 0) a1 = stem_a1;
        else if (fuse_c1) {       // Bottleneck conv1 (1x1 64->32 + ReLU) rides in the epilogue of stem conv2
            const int pc1 = b.pack(b.next(32, 64, 1, 1), nullptr, 0);
            a1 = b.new_tensor(32, H, W);
            net->ops[stem2_op].conv2 = pc1; net->ops[stem2_op].dst2 = a1;
        } else a1 = b.conv(x, 32, 1, 1, 1);
        const int a2 = b.conv(a1, 32, 3, 1, 1);
        const FoldedConv& c3 = b.next(128, 32, 1, 1);
        const FoldedConv& ds = b.next(128, 64, 1, 1);
        const int pc = b.pack(c3, &ds, 0);
        const bool fuse = net->dtype == TTUP_DTYPE_BF16 && !getenv("TTUP_NO_FUSE") && H % 2 == 0 && W % 2 == 0;
        if (fuse) {
            const int p5 = b.pack(b.next(16, 128, 3, 1), nullptr, 0);
            const int p6 = b.pack(b.next(32, 128, 3, 2), nullptr, 0);
            xs[0] = b.new_tensor(16, H, W);
            xs[1] = b.new_tensor(32, H / 2, W / 2);
            Op op; op.kind = Op::BNECK_TRANS; op.conv = pc; op.conv2 = p5; op.conv3 = p6; op.src0 = a2; op.src1 = x; op.dst = xs[0]; op.dst2 = xs[1];
            net->ops.push_back(op);
        } else {
            const int dst = b.new_tensor(128, H, W);
            Op op; op.conv = pc; op.src0 = a2; op.src1 = x; op.dst = dst; op.relu = 1; net->ops.push_back(op);
            x = dst; net->taps["layer1"] = x;
            xs[0] = b.conv(x, 16, 3, 1, 1);
            xs[1] = b.conv(x, 32, 3, 2, 1);
        }
        net->taps["trans1_0"] = xs[0]; net->taps["trans1_1"] = xs[1];
    }
    std::vector<int> ys = b.stage(xs, 2);
    net->taps["stage2_0"] = ys[0]; net->taps["stage2_1"] = ys[1];
    // transition2: new branch from the last output (wasb.py:462-467)
    xs = {ys[0], ys[1], b.conv(ys[1], 64, 3, 2, 1)};
    ys = b.stage(xs, 3);
    net->taps["stage3_0"] = ys[0]; net->taps["stage3_1"] = ys[1]; net->taps["stage3_2"] = ys[2];
    xs = {ys[0], ys[1], ys[2], b.conv(ys[2], 128, 3, 2, 1)};
    const bool head_in_chain = net->dtype == TTUP_DTYPE_BF16 && net->n_out == 1 && !getenv("TTUP_NO_FUSE") && !getenv("TTUP_NO_FUSE_SUM");
    ys = b.stage(xs, 1, head_in_chain);
    net->t_out = ys[0];
    if (head_in_chain && net->ops.back().kind == Op::BB_CHAIN && net->ops.back().head) {
        net->fused_head = true;                       // stage-4 output 0 lives only in the registers of the last block chain
    } else if (net->dtype == TTUP_DTYPE_BF16 && net->n_out == 1 && !getenv("TTUP_NO_FUSE") && net->ops.back().kind == Op::UPSUM && net->ops.back().dst == ys[0]) {
        net->ops.back().kind = Op::UPSUM_HEAD;        // stage-4 output 0 is consumed in registers and never stored
        net->fused_head = true;
    } else {
        net->taps["stage4_0"] = ys[0];
    }
    if (b.rc) return b.rc;
    TTUP_REQUIRE(b.cursor == folded.size() - 1, TTUP_EFORMAT, "wasb: consumed %zu of %zu convs", b.cursor, folded.size() - 1);
    return TTUP_OK;
}

int run_op(ttup_wasb* net, const Op& op, int mb, hipStream_t st) {
    {
        if (op.kind == Op::CONV) {
            const Tensor& s = net->tensors[op.src0];
            ConvLaunch l;
            l.src0 = s.ptr; l.src1 = op.src1 >= 0 ? net->tensors[op.src1].ptr : nullptr;
            l.residual = op.residual >= 0 ? net->tensors[op.residual].ptr : nullptr;
            l.dst = net->tensors[op.dst].ptr; l.batch = mb; l.h = s.h; l.w = s.w; l.relu = op.relu; l.n_active = net->n_active;
            if (!net->op_roi.empty() && net->roi_flag) { l.roi = net->op_roi[&op - net->ops.data()]; l.roi.flag = net->roi_flag; }
            if (op.conv2 >= 0) { l.follow = &net->convs[op.conv2]; l.dst2 = net->tensors[op.dst2].ptr; }
            if (op.lin16 >= 0) { l.lin16 = &net->convs[op.lin16]; l.lin16_dst = net->tensors[op.lin16_dst].ptr; }
            if (op.lin32 >= 0) { l.lin32 = &net->convs[op.lin32]; l.lin32_dst = net->tensors[op.lin32_dst].ptr; }
            if (op.pair >= 0) { l.pair = &net->convs[op.pair]; l.pair_dst = net->tensors[op.pair_dst].ptr; l.pair_relu = op.pair_relu; }
            if (op.res2 >= 0) l.res2 = net->tensors[op.res2].ptr;
            if (op.res3 >= 0) { l.res3 = net->tensors[op.res3].ptr; l.sh3 = op.sh3; }
            const int rc = launch_conv(net->convs[op.conv], l, net->dtype, st);
            if (rc) return rc;
        } else if (op.kind == Op::STEM) {
            const Tensor& s = net->tensors[op.src0];
            const bool fm = net->frames_mode && net->t_frames >= 0 && op.conv1f >= 0;
            const int rc = launch_stem(net->convs[fm ? op.conv1f : op.conv], net->convs[op.conv2], net->convs[op.conv3], fm ? net->tensors[net->t_frames].ptr : s.ptr,
                                       net->tensors[op.dst].ptr, net->tensors[op.dst2].ptr, mb, s.h, s.w, st, fm ? net->in_ch / 3 : 0);
            if (rc) return rc;
        } else if (op.kind == Op::UPSUM_HEAD) {
            return TTUP_OK;       // launched by forward_micro (run_head_op), which knows the output buffers
        } else if (op.kind == Op::BB_CHAIN) {
            if (op.head) return TTUP_OK;      // launched by forward_micro (run_head_op), which knows the output buffers
            const Tensor& s = net->tensors[op.src0];
            const PackedConv* cv[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < op.n_chain; ++k) cv[k] = &net->convs[op.chain[k]];
            int rc;
            if (op.n_chain == 4 && op.n_terms > 0) {          // fuse-layer sum in the epilogue: dst2 = relu(dst + sum up(terms))
                BBSum sum;
                sum.n_terms = op.n_terms;
                for (int k = 0; k < op.n_terms; ++k) { sum.terms[k] = net->tensors[op.terms[k]].ptr; sum.shifts[k] = op.shifts[k]; }
                sum.ysum = net->tensors[op.dst2].ptr;
                rc = launch_bb_chain(cv, 4, s.ptr, op.dst >= 0 ? net->tensors[op.dst].ptr : nullptr, mb, s.h, s.w, nullptr, nullptr, st, &sum);
            } else {
                rc = launch_bb_chain(cv, op.n_chain, s.ptr, net->tensors[op.dst].ptr, mb, s.h, s.w,
                                     op.conv2 >= 0 ? &net->convs[op.conv2] : nullptr, op.dst2 >= 0 ? net->tensors[op.dst2].ptr : nullptr, st);
            }
            if (rc) return rc;
        } else if (op.kind == Op::BNECK_TRANS) {
            const Tensor& s = net->tensors[op.src0];
            const int rc = launch_bneck_trans(net->convs[op.conv], net->convs[op.conv2], net->convs[op.conv3], s.ptr, net->tensors[op.src1].ptr,
                                              net->tensors[op.dst].ptr, net->tensors[op.dst2].ptr, mb, s.h, s.w, st);
            if (rc) return rc;
        } else {
            const Tensor& d = net->tensors[op.dst];
            const void* terms[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < op.n_terms; ++k) terms[k] = net->tensors[op.terms[k]].ptr;
            Roi roi;
            if (!net->op_roi.empty() && net->roi_flag) { roi = net->op_roi[&op - net->ops.data()]; roi.flag = net->roi_flag; }
            const int rc = launch_upsum(net->tensors[op.src0].ptr, terms, op.shifts, op.n_terms, d.ptr, mb, d.h, d.w, d.c, net->dtype, st, net->n_active, &roi);
            if (rc) return rc;
        }
    }
    return TTUP_OK;
}

int run_graph(ttup_wasb* net, int mb, hipStream_t st) {
    for (const Op& op : net->ops) { const int rc = run_op(net, op, mb, st); if (rc) return rc; }
    return TTUP_OK;
}

// the fused last op of the bf16 ball path: stage-4 fuse sum + head + argmax partials (+ window gather)
int run_head_op(ttup_wasb* net, int mb, float* heat, long long* am, float* wn, hipStream_t st) {
    const Op& op = net->ops.back();
    if (op.kind == Op::BB_CHAIN) {
        // last block chain of the full-resolution branch + stage-4 fuse sum + 1x1 head + per-tile argmax partials
        const Tensor& s = net->tensors[op.src0];
        const PackedConv* cv[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < op.n_chain; ++k) cv[k] = &net->convs[op.chain[k]];
        const int nblk = bb_chain_tiles_per_img(s.h, s.w);
        TTUP_REQUIRE(net->refine_ws && net->refine_ws_bytes >= (size_t)mb * nblk * 12, TTUP_EINVAL, "head: workspace too small");
        BBSum sum;
        sum.n_terms = op.n_terms;
        for (int k = 0; k < op.n_terms; ++k) { sum.terms[k] = net->tensors[op.terms[k]].ptr; sum.shifts[k] = op.shifts[k]; }
        sum.heat = heat; sum.head_w = net->head_w_dev; sum.head_bias = net->head_bias;
        sum.pi = (long long*)net->refine_ws; sum.pv = (float*)(sum.pi + (size_t)mb * nblk);
        int rc = launch_bb_chain(cv, 4, s.ptr, nullptr, mb, s.h, s.w, nullptr, nullptr, st, &sum);
        if (rc) return rc;
        if (am || wn) {
            TTUP_REQUIRE(am && wn, TTUP_EINVAL, "head: argmax and window outputs come together");
            rc = launch_argmax_finish(heat, mb, s.h, s.w, nblk, sum.pv, sum.pi, am, wn, st);
        }
        return rc;
    }
    const void* terms[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < op.n_terms; ++k) terms[k] = net->tensors[op.terms[k]].ptr;
    return launch_upsum_head(net->tensors[op.src0].ptr, terms, op.shifts, op.n_terms, net->head_w_dev, net->head_bias, heat, mb, net->H, net->W,
                             am, wn, net->refine_ws, net->refine_ws_bytes, st);
}

int forward_micro(ttup_wasb* net, const float* x_dev, const uint8_t* frames_dev, int n_frames, int src_h, int src_w, int batch, int b0,
                  float* heat_dev, int64_t* argmax_dev, float* win_dev, int lane, hipStream_t st) {
    const int H = net->H, W = net->W;
    const size_t hw = (size_t)H * W;
    const int K = net->n_out;
    const int mb = batch - b0 < net->micro ? batch - b0 : net->micro;
    net->use_lane(lane);
    int rc;
    const int nf = net->in_ch / 3;
    net->frames_mode = !x_dev && net->t_frames >= 0;
    if (x_dev) rc = launch_nchw_to_nhwc(x_dev + (size_t)b0 * net->in_ch * hw, net->tensors[net->t_input].ptr, mb, net->in_ch, 16, H, W, net->dtype, st);
    else if (net->frames_mode)      // every frame of the micro-batch (mb + nf - 1 of them) is pre-processed once; the stem assembles the samples
        rc = launch_preprocess(frames_dev, n_frames, src_h, src_w, H, W, net->tensors[net->t_frames].ptr, TTUP_LAYOUT_NHWC4_FRAME, net->dtype, b0, mb + nf - 1, 1, st);
    else rc = launch_preprocess(frames_dev, n_frames, src_h, src_w, H, W, net->tensors[net->t_input].ptr, TTUP_LAYOUT_NHWC16, net->dtype, b0, mb, nf, st);
    if (rc) return rc;
    rc = run_graph(net, mb, st);
    if (rc) return rc;
    float* heat = heat_dev ? heat_dev + (size_t)b0 * K * hw : net->heat_scratch;
    if (net->fused_head) {
        const bool peaks = argmax_dev || win_dev;
        long long* am = peaks ? (argmax_dev ? (long long*)argmax_dev + b0 : net->argmax_scratch) : nullptr;
        float* wn = peaks ? (win_dev ? win_dev + (size_t)b0 * 9 : net->win_scratch) : nullptr;
        rc = run_head_op(net, mb, heat, am, wn, st);
        if (rc) return rc;
        // certified argmax: pixels within 2*eps of this bf16 maximum are the only ones that can be the fp32 argmax
        if (net->cert.enabled && argmax_dev && win_dev) return cert_scan(net, heat, am, b0, mb, st);
        return TTUP_OK;
    }
    rc = launch_head(net->tensors[net->t_out].ptr, net->head_w_dev, net->head_b_dev, K, heat, mb, H, W, 16, net->dtype, st, net->n_active);
    if (rc) return rc;
    if (argmax_dev || win_dev) {
        long long* am = argmax_dev ? (long long*)argmax_dev + (size_t)b0 * K : net->argmax_scratch;
        float* wn = win_dev ? win_dev + (size_t)b0 * K * 9 : net->win_scratch;
        rc = refine_argmax(heat, mb * K, H, W, am, wn, net->refine_ws, net->refine_ws_bytes, st);
        if (rc) return rc;
        // certified argmax of the multi-channel head (table keypoints): the same scan / plan per heatmap, crops shared by a frame's channels
        if (net->cert.enabled && argmax_dev && win_dev) return cert_scan(net, heat, am, b0, mb, st);
    }
    return TTUP_OK;
}

int forward_impl(ttup_wasb* net, const float* x_dev, const uint8_t* frames_dev, int n_frames, int src_h, int src_w,
                 int batch, float* heat_dev, int64_t* argmax_dev, float* win_dev, hipStream_t st) {
    const int n_micro = (batch + net->micro - 1) / net->micro;
    const int n_lanes = n_micro < (int)net->lanes.size() ? (n_micro > 0 ? n_micro : 1) : (int)net->lanes.size();
    const hipStream_t caller = st;
    const bool certify = net->cert.enabled && argmax_dev && win_dev && batch > 0;
    // Consecutive calls may come in on different caller streams (StreamWorker.submit alternates two) while the activations, the
    // heatmap scratch and the argmax workspace of a lane belong to ONE micro-batch at a time.  Handles with lane streams run every
    // micro-batch on its lane's stream -- also when the call has a single micro-batch -- so stream order serialises the lane's
    // buffers.  Single-lane handles (max_batch <= micro-batch, TTUP_LANES=1) run on the caller's stream: a call then waits for the
    // previous call's last micro-batch, whatever stream that call was issued on.
    const bool lane_streams = net->lanes[0].stream != nullptr;
    if (!lane_streams && net->pass_recorded) TTUP_HIP_CHECK(hipStreamWaitEvent(caller, net->pass_done, 0));
    if (certify) { const int rc = cert_begin(net, batch, caller); if (rc) return rc; }
    if (lane_streams) {
        TTUP_HIP_CHECK(hipEventRecord(net->fork, caller));
        for (int l = 0; l < n_lanes; ++l) TTUP_HIP_CHECK(hipStreamWaitEvent(net->lanes[l].stream, net->fork, 0));
    }
    int rc_all = TTUP_OK, last_lane = 0;
    for (int b0 = 0, i = 0; b0 < batch && rc_all == TTUP_OK; b0 += net->micro, ++i) {
        rc_all = forward_micro(net, x_dev, frames_dev, n_frames, src_h, src_w, batch, b0, heat_dev, argmax_dev, win_dev,
                               i % n_lanes, lane_streams ? net->lanes[i % n_lanes].stream : caller);
        last_lane = i % n_lanes;
    }
    if (lane_streams) {       // join even after an error so the caller's stream never runs ahead of enqueued work
        for (int l = 0; l < n_lanes; ++l) {
            (void)hipEventRecord(net->lanes[l].done, net->lanes[l].stream);
            (void)hipStreamWaitEvent(caller, net->lanes[l].done, 0);
        }
    } else if (hipEventRecord(net->pass_done, caller) == hipSuccess) net->pass_recorded = true;
    net->use_lane(last_lane);
    if (rc_all) return rc_all;
    net->last_batch = batch < net->micro ? batch : net->micro;
    if (certify) return cert_finish(net, x_dev, frames_dev, n_frames, src_h, src_w, batch, argmax_dev, win_dev, caller);
    return TTUP_OK;
}

}  // namespace

namespace ttup {
int run_ops(ttup_wasb* net, int mb, hipStream_t st) { return run_graph(net, mb, st); }

// Cone pruning of a layer-by-layer fp32 graph: walk the ops backwards from the heatmap region [lo, hi) x [lo, hi) and record, for
// every tensor, the union of what its consumers read.  A 3x3 conv reads one pixel around its outputs (times the stride), a fuse sum
// reads the same pixels of its base and pixel >> shift of every upsampled term.  The recorded regions are SUPERSETS by construction
// (kernels round them out to whole tiles): every value an op reads inside its own region has been produced.
int compute_roi(ttup_wasb* net, int lo, int hi, int lo2, int hi2) {
    TTUP_REQUIRE(net && net->dtype == TTUP_DTYPE_F32 && net->t_out >= 0, TTUP_EINVAL, "compute_roi: an fp32 handle is expected");
    const size_t nt = net->tensors.size();
    struct R { int y0, y1, x0, x1; bool any; };
    // one backward walk from the heatmap region [l, h) x [l, h): out[k] = the region of op k's output in the cone (any == false: none)
    auto walk = [&](int l, int h, std::vector<R>& out) -> int {
        std::vector<R> need(nt, R{0, 0, 0, 0, false});
        auto add = [&](int t, int y0, int y1, int x0, int x1) {
            const Tensor& tn = net->tensors[t];
            y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0; y1 = y1 > tn.h ? tn.h : y1; x1 = x1 > tn.w ? tn.w : x1;
            if (y1 <= y0 || x1 <= x0) return;
            R& r = need[t];
            if (!r.any) r = R{y0, y1, x0, x1, true};
            else { r.y0 = y0 < r.y0 ? y0 : r.y0; r.y1 = y1 > r.y1 ? y1 : r.y1; r.x0 = x0 < r.x0 ? x0 : r.x0; r.x1 = x1 > r.x1 ? x1 : r.x1; }
        };
        add(net->t_out, l, h, l, h);
        out.assign(net->ops.size(), R{0, 1, 0, 1, false});          // nothing in the cone reads the op: one pixel
        for (int k = (int)net->ops.size() - 1; k >= 0; --k) {
            const Op& op = net->ops[k];
            TTUP_REQUIRE(op.kind == Op::CONV || op.kind == Op::UPSUM, TTUP_EINVAL, "compute_roi: fused op in an fp32 graph");
            TTUP_REQUIRE(op.conv2 < 0 && op.lin16 < 0 && op.lin32 < 0 && op.pair < 0 && op.res2 < 0 && op.res3 < 0, TTUP_EINVAL, "compute_roi: fused epilogue in an fp32 graph");
            const R d = need[op.dst];
            if (!d.any) continue;
            out[k] = d;
            if (op.kind == Op::CONV) {
                const PackedConv& p = net->convs[op.conv];
                const int s_ = p.stride, pad = p.k / 2;
                add(op.src0, d.y0 * s_ - pad, (d.y1 - 1) * s_ + pad + 1, d.x0 * s_ - pad, (d.x1 - 1) * s_ + pad + 1);
                if (op.src1 >= 0) add(op.src1, d.y0 * s_ - pad, (d.y1 - 1) * s_ + pad + 1, d.x0 * s_ - pad, (d.x1 - 1) * s_ + pad + 1);
                if (op.residual >= 0) add(op.residual, d.y0, d.y1, d.x0, d.x1);
            } else {
                add(op.src0, d.y0, d.y1, d.x0, d.x1);
                for (int j = 0; j < op.n_terms; ++j) { const int sh = op.shifts[j]; add(op.terms[j], d.y0 >> sh, ((d.y1 - 1) >> sh) + 1, d.x0 >> sh, ((d.x1 - 1) >> sh) + 1); }
            }
        }
        return TTUP_OK;
    };
    std::vector<R> r1, r2;
    if (int rc = walk(lo, hi, r1)) return rc;
    const bool two = hi2 > lo2;
    if (two) { if (int rc = walk(lo2, hi2, r2)) return rc; }
    net->op_roi.assign(net->ops.size(), Roi());
    for (size_t k = 0; k < net->ops.size(); ++k) {
        Roi& o = net->op_roi[k];
        o.y0 = r1[k].y0; o.y1 = r1[k].y1; o.x0 = r1[k].x0; o.x1 = r1[k].x1;
        if (two) { o.sy0 = r2[k].y0; o.sy1 = r2[k].y1; o.sx0 = r2[k].x0; o.sx1 = r2[k].x1; }
    }
    if (getenv("TTUP_DEBUG_ROI")) {
        double full = 0, kept = 0, kept2 = 0;
        for (size_t k = 0; k < net->ops.size(); ++k) {
            const Op& op = net->ops[k];
            const Tensor& d = net->tensors[op.dst];
            const Roi& o = net->op_roi[k];
            double w = (double)d.h * d.w;
            if (op.kind == Op::CONV) { const PackedConv& p = net->convs[op.conv]; w *= (double)p.cout * p.cin_total * p.k * p.k; } else w *= d.c;
            full += w; kept += w * ((double)(o.y1 - o.y0) * (o.x1 - o.x0)) / ((double)d.h * d.w);
            kept2 += w * ((double)(o.sy1 - o.sy0) * (o.sx1 - o.sx0)) / ((double)d.h * d.w);
            fprintf(stderr, "roi op %2zu %s dst %3dx%3dx%3d -> [%d,%d)x[%d,%d)  class 2 [%d,%d)x[%d,%d)\n", k, op.kind == Op::CONV ? "conv " : "upsum", d.h, d.w, d.c, o.y0, o.y1, o.x0, o.x1,
                    o.sy0, o.sy1, o.sx0, o.sx1);
        }
        fprintf(stderr, "roi: %.1f %% of the graph's multiply-adds kept (class 2: %.1f %%)\n", 100.0 * kept / full, 100.0 * kept2 / full);
    }
    net->out_roi = Roi();
    net->out_roi.y0 = lo; net->out_roi.y1 = hi; net->out_roi.x0 = lo; net->out_roi.x1 = hi;
    if (two) { net->out_roi.sy0 = lo2; net->out_roi.sy1 = hi2; net->out_roi.sx0 = lo2; net->out_roi.sx1 = hi2; }
    return TTUP_OK;
}
}

// micro_override / lanes_override > 0 fix the micro-batch and the lane count (the certified argmax's fp32 crop net runs its
// whole batch as one micro-batch on one lane); 0 = TTUP_MICRO_BATCH / TTUP_LANES or the defaults
int ttup_wasb_create_internal(const void* blob, size_t blob_bytes, int height, int width, int max_batch, int dtype, int micro_override, int lanes_override, ttup_wasb** out) {
    TTUP_REQUIRE(blob && out, TTUP_EINVAL, "ttup_wasb_create: null pointer");
    TTUP_REQUIRE(height > 0 && width > 0 && height % 8 == 0 && width % 8 == 0, TTUP_EINVAL,
                 "ttup_wasb_create: input size %dx%d must be positive multiples of 8", height, width);
    TTUP_REQUIRE(max_batch > 0, TTUP_EINVAL, "ttup_wasb_create: max_batch must be positive");
    TTUP_REQUIRE(dtype == TTUP_DTYPE_BF16 || dtype == TTUP_DTYPE_F32, TTUP_EINVAL, "ttup_wasb_create: unknown dtype %d", dtype);
    int ndev = 0;
    TTUP_HIP_CHECK(hipGetDeviceCount(&ndev));
    TTUP_REQUIRE(ndev > 0, TTUP_EHIP, "ttup_wasb_create: no HIP device");
    std::vector<FoldedConv> folded;
    std::vector<float> head_w, head_b;
    int in_ch = 0, head_out = 0;
    int rc = parse_blob(blob, blob_bytes, &folded, &in_ch, &head_out, &head_w, &head_b);
    if (rc) return rc;
    TTUP_REQUIRE((in_ch == 9 || in_ch == 3) && head_out >= 1 && head_out <= 16, TTUP_EFORMAT, "wasb blob: in_ch=%d head_out=%d unsupported (ball detector 9/3, table detector 3/13)", in_ch, head_out);
    std::unique_ptr<ttup_wasb> net(new ttup_wasb);
    net->H = height; net->W = width; net->max_batch = max_batch; net->dtype = dtype; net->in_ch = in_ch;
    net->blob.assign((const char*)blob, (const char*)blob + blob_bytes);
    net->n_out = head_out == 3 ? 1 : head_out;       // ball detector keeps the middle of its 3 channels (wasb.py:606)
    // micro-batch: enough tiles to fill 256 CUs, small enough that layer outputs stay cache-friendly
    const char* env = getenv("TTUP_MICRO_BATCH");
    int micro = micro_override > 0 ? micro_override : (env ? atoi(env) : 8);
    if (micro < 1) micro = 1;
    net->micro = micro < max_batch ? micro : max_batch;
    rc = build(net.get(), folded);
    if (rc) return rc;
    // head weights of the returned channels: the ball detector keeps only channel 1 of its 3 (wasb.py:606), the table
    // detector all 13 (tabledetection/models/hrnet.py:586-589)
    {
        const int first = head_out == 3 ? 1 : 0;
        TTUP_HIP_CHECK(hipMalloc((void**)&net->head_w_dev, (size_t)net->n_out * 16 * sizeof(float)));
        TTUP_HIP_CHECK(hipMemcpy(net->head_w_dev, head_w.data() + (size_t)first * 16, (size_t)net->n_out * 16 * sizeof(float), hipMemcpyHostToDevice));
        TTUP_HIP_CHECK(hipMalloc((void**)&net->head_b_dev, (size_t)net->n_out * sizeof(float)));
        TTUP_HIP_CHECK(hipMemcpy(net->head_b_dev, head_b.data() + first, (size_t)net->n_out * sizeof(float), hipMemcpyHostToDevice));
        net->head_bias = head_b[first];
    }
    const size_t hw = (size_t)height * width;
    net->refine_ws_bytes = ttup_refine_workspace_bytes(net->micro * net->n_out, height, width);
    if (upsum_head_ws_bytes(net->micro, height, width) > net->refine_ws_bytes) net->refine_ws_bytes = upsum_head_ws_bytes(net->micro, height, width);
    {
        const char* le = getenv("TTUP_LANES");
        int n_lanes = lanes_override > 0 ? lanes_override : (le ? atoi(le) : 2);
        const int n_micro = (max_batch + net->micro - 1) / net->micro;
        if (n_lanes > n_micro) n_lanes = n_micro;
        if (n_lanes < 1) n_lanes = 1;
        if (n_lanes > 4) n_lanes = 4;
        net->lanes.resize(n_lanes);
        for (int l = 0; l < n_lanes; ++l) {
            ttup_wasb::Lane& L = net->lanes[l];
            L.ptr.assign(net->tensors.size(), nullptr);
            for (size_t i = 0; i < net->tensors.size(); ++i) {
                const Tensor& t = net->tensors[i];
                if (l == 0) { L.ptr[i] = t.ptr; continue; }      // lane 0 adopts the buffers the builder allocated
                TTUP_HIP_CHECK(hipMalloc(&L.ptr[i], (size_t)(net->micro + t.extra) * t.h * t.w * t.c * net->esize()));
            }
            TTUP_HIP_CHECK(hipMalloc((void**)&L.heat_scratch, (size_t)net->micro * net->n_out * hw * sizeof(float)));
            TTUP_HIP_CHECK(hipMalloc(&L.refine_ws, net->refine_ws_bytes));
            TTUP_HIP_CHECK(hipMalloc((void**)&L.argmax_scratch, (size_t)net->micro * net->n_out * sizeof(long long)));
            TTUP_HIP_CHECK(hipMalloc((void**)&L.win_scratch, (size_t)net->micro * net->n_out * 9 * sizeof(float)));
            if (n_lanes > 1) {
                TTUP_HIP_CHECK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
                TTUP_HIP_CHECK(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
            }
        }
        if (n_lanes > 1) TTUP_HIP_CHECK(hipEventCreateWithFlags(&net->fork, hipEventDisableTiming));
        TTUP_HIP_CHECK(hipEventCreateWithFlags(&net->pass_done, hipEventDisableTiming));
        net->use_lane(0);
    }
    // measurement aid (tools/ops_report_f32.py): TTUP_DEBUG_FORCE_ROI=1 prunes EVERY sample of an fp32 handle to the cone of its central
    // 24-pixel core, as the certified argmax's crop net does for interior crops -- per-op timings of the pruned graph (the flags leak
    // with the process: a debugging switch)
    if (dtype == TTUP_DTYPE_F32 && getenv("TTUP_DEBUG_FORCE_ROI") && net->H == net->W && net->H >= 2 * 72 + 24) {
        int* flags = nullptr;
        TTUP_HIP_CHECK(hipMalloc((void**)&flags, (size_t)max_batch * sizeof(int)));
        std::vector<int> ones((size_t)max_batch, atoi(getenv("TTUP_DEBUG_FORCE_ROI")) == 2 ? 2 : 1);          // = 2: the class-2 regions (16-pixel core)
        TTUP_HIP_CHECK(hipMemcpy(flags, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice));
        if (int rc = compute_roi(net.get(), 72, net->H - 72, 72, 88)) return rc;
        net->roi_flag = flags;
    }
    TTUP_HIP_CHECK(hipDeviceSynchronize());
    *out = net.release();
    return TTUP_OK;
}

extern "C" int ttup_wasb_create(const void* blob, size_t blob_bytes, int height, int width, int max_batch, int dtype, ttup_wasb** out) {
    return ttup_wasb_create_internal(blob, blob_bytes, height, width, max_batch, dtype, 0, 0, out);
}

// micro_batch / lanes: 0 = the defaults (8 frames, two lanes; TTUP_MICRO_BATCH / TTUP_LANES); lanes = 1 runs every micro-batch on
// the caller's stream -- for a handle that shares the device with another busy handle (the hub pipeline's two detectors), where a
// second lane only adds streams that collide on the runtime's few hardware queues
extern "C" int ttup_wasb_create_ex(const void* blob, size_t blob_bytes, int height, int width, int max_batch, int dtype, int micro_batch, int lanes,
                                   ttup_wasb** out) {
    TTUP_REQUIRE(micro_batch >= 0 && lanes >= 0 && lanes <= 4, TTUP_EINVAL, "ttup_wasb_create_ex: micro_batch %d / lanes %d out of range", micro_batch, lanes);
    return ttup_wasb_create_internal(blob, blob_bytes, height, width, max_batch, dtype, micro_batch, lanes, out);
}

// The handle's internal streams: its lane streams (none for a single-lane handle), then the stream of the certified argmax's fp32
// passes (when enabled).  For callers that want to know which streams share a hardware queue (tools/queue_probe.py).
extern "C" int ttup_wasb_streams(ttup_wasb* net, void** out, int cap, int* n_out) {
    TTUP_REQUIRE(net && out && n_out && cap >= 0, TTUP_EINVAL, "ttup_wasb_streams: bad argument");
    int n = 0;
    for (auto& L : net->lanes) if (L.stream && n < cap) out[n++] = (void*)L.stream;
    if (net->cert.enabled && net->cert.stream && n < cap) out[n++] = (void*)net->cert.stream;
    *n_out = n;
    return TTUP_OK;
}

extern "C" void ttup_wasb_destroy(ttup_wasb* net) {
    if (!net) return;
    (void)hipDeviceSynchronize();
    delete net;
}

extern "C" int ttup_wasb_forward(ttup_wasb* net, const float* x_dev, int batch, float* heat_dev, int64_t* argmax_dev, float* win_dev, void* stream) {
    TTUP_REQUIRE(net && x_dev, TTUP_EINVAL, "ttup_wasb_forward: null pointer");
    TTUP_REQUIRE(batch >= 0 && batch <= net->max_batch, TTUP_EINVAL, "ttup_wasb_forward: batch %d outside [0,%d]", batch, net->max_batch);
    return forward_impl(net, x_dev, nullptr, 0, 0, 0, batch, heat_dev, argmax_dev, win_dev, (hipStream_t)stream);
}

extern "C" int ttup_wasb_forward_frames(ttup_wasb* net, const uint8_t* frames_dev, int n_frames, int src_h, int src_w,
                                        float* heat_dev, int64_t* argmax_dev, float* win_dev, void* stream) {
    TTUP_REQUIRE(net && frames_dev, TTUP_EINVAL, "ttup_wasb_forward_frames: null pointer");
    const int nf = net->in_ch / 3;       // 3 frames per sample for the ball detector, 1 for the table detector
    TTUP_REQUIRE(n_frames >= nf && src_h > 0 && src_w > 0, TTUP_EINVAL, "ttup_wasb_forward_frames: need at least %d frames", nf);
    const int batch = n_frames - (nf - 1);
    TTUP_REQUIRE(batch <= net->max_batch, TTUP_EINVAL, "ttup_wasb_forward_frames: %d triples exceed max_batch %d", batch, net->max_batch);
    return forward_impl(net, nullptr, frames_dev, n_frames, src_h, src_w, batch, heat_dev, argmax_dev, win_dev, (hipStream_t)stream);
}

extern "C" int ttup_wasb_read_tap(ttup_wasb* net, const char* name, int batch, float* out_dev, int* c, int* h, int* w, void* stream) {
    TTUP_REQUIRE(net && name, TTUP_EINVAL, "ttup_wasb_read_tap: null pointer");
    auto it = net->taps.find(name);
    TTUP_REQUIRE(it != net->taps.end(), TTUP_EINVAL, "ttup_wasb_read_tap: unknown tap '%s'", name);
    const Tensor& t = net->tensors[it->second];
    if (c) *c = t.c; if (h) *h = t.h; if (w) *w = t.w;
    if (!out_dev) return TTUP_OK;
    TTUP_REQUIRE(batch > 0 && batch <= net->micro, TTUP_EINVAL, "ttup_wasb_read_tap: batch %d exceeds the micro-batch %d held in memory", batch, net->micro);
    return launch_nhwc_to_nchw(t.ptr, out_dev, batch, t.c, t.h, t.w, net->dtype, (hipStream_t)stream);
}

// ---- measurement aids for bench.py
// info: 8 ints per op {kind(0 conv,1 upsum,2 bneck_trans,3 bb_chain,4 stem,5 upsum_head), algorithmic MACs per output element
// (or cin), cout, k, stride, out_h, out_w, cin_padded / chain length}; name: the HIP kernel the op launches.
namespace {
void op_info(const ttup_wasb* net, int i, int* o, char* name) {
    const Op& op = net->ops[i];
    const Tensor& d = net->tensors[op.dst >= 0 ? op.dst : op.src0];
    const bool bf = net->dtype == TTUP_DTYPE_BF16;
    char nm[64] = "";
    if (op.kind == Op::STEM) {
        o[0] = 4; o[1] = 9 * 9 * 64 + 9 * 64 * 64 + 64 * 32; o[2] = 1; o[3] = 1; o[4] = 1; o[5] = d.h; o[6] = d.w; o[7] = 0;
        snprintf(nm, sizeof nm, "stem_kernel");
    } else if (op.kind == Op::BB_CHAIN) {
        o[0] = 3; o[1] = op.n_chain * d.c * 9; o[2] = d.c; o[3] = 1; o[4] = 1; o[5] = d.h; o[6] = d.w; o[7] = op.n_chain;
        if (op.n_chain == 4) snprintf(nm, sizeof nm, "bb_chain2_kernel<16>%s", op.head ? "+sum+head" : op.n_terms > 0 ? "+sum" : "");
        else snprintf(nm, sizeof nm, "bb_chain_kernel<%d,1>%s", d.c, op.conv2 >= 0 ? "+1x1" : "");
        if (op.conv2 >= 0) o[1] += 16;          // fused 1x1 32->16 follower: 32*16 MACs per pixel = 16 per output element of the block
    } else if (op.kind == Op::BNECK_TRANS) {
        // algorithmic MACs per output pixel of B0: 96*128 (1x1) + 1152*16 (3x3 s1) + 1152*32/4 (3x3 s2 at quarter density)
        o[0] = 2; o[1] = 96 * 128 + 1152 * 16 + 1152 * 8; o[2] = 1; o[3] = 1; o[4] = 1; o[5] = d.h; o[6] = d.w; o[7] = 0;
        snprintf(nm, sizeof nm, "bneck_trans_kernel");
    } else if (op.kind == Op::CONV) {
        const PackedConv& pc = net->convs[op.conv];
        o[0] = 0; o[1] = (i == 0) ? net->in_ch : pc.cin_total; o[2] = pc.cout; o[3] = pc.k; o[4] = pc.stride; o[5] = d.h; o[6] = d.w; o[7] = pc.cin_total;
        if (!bf) snprintf(nm, sizeof nm, "conv_direct_f32_kernel");
        else if (pc.k == 3 && pc.stride == 1 && pc.cout == 64 && pc.cin_total == 64 && op.conv2 < 0) snprintf(nm, sizeof nm, "conv64_kernel%s", (op.lin16 >= 0 || op.lin32 >= 0) ? "+1x1" : "");
        else if (op.pair >= 0) { snprintf(nm, sizeof nm, "conv_s2_pair_kernel"); o[2] = pc.cout + net->convs[op.pair].cout; }      // both convs' outputs count
        else snprintf(nm, sizeof nm, "conv_mfma_kernel<%d,%d,%d,%d>", pc.ck, pc.cout, pc.k, pc.stride);
    } else {
        o[0] = op.kind == Op::UPSUM_HEAD ? 5 : 1; o[1] = op.n_terms; o[2] = d.c; o[3] = 0; o[4] = 0; o[5] = d.h; o[6] = d.w; o[7] = d.c;
        snprintf(nm, sizeof nm, op.kind == Op::UPSUM_HEAD ? "upsum_head_kernel" : bf ? "upsum_bf16x8_kernel" : "upsum_kernel<float>");
    }
    if (name) { memset(name, 0, 64); memcpy(name, nm, strlen(nm)); }
}
}  // namespace

// Every op on its own: `reps` back-to-back launches of one op between two HIP events on `stream` (inputs warm in the caches).
extern "C" int ttup_wasb_time_ops(ttup_wasb* net, int batch, int reps, int max_ops, float* ms_out, int* info_out, int* n_ops_out, void* stream) {
    TTUP_REQUIRE(net && ms_out && info_out && n_ops_out, TTUP_EINVAL, "ttup_wasb_time_ops: null pointer");
    TTUP_REQUIRE(batch > 0 && batch <= net->micro && reps > 0, TTUP_EINVAL, "ttup_wasb_time_ops: batch must be in [1,%d]", net->micro);
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)net->ops.size();
    TTUP_REQUIRE(n <= max_ops, TTUP_EINVAL, "ttup_wasb_time_ops: %d ops exceed max_ops %d", n, max_ops);
    hipEvent_t e0, e1;
    TTUP_HIP_CHECK(hipEventCreate(&e0));
    TTUP_HIP_CHECK(hipEventCreate(&e1));
    int rc = TTUP_OK;
    net->use_lane(0);
    static const int only_op = getenv("TTUP_TIME_ONLY_OP") ? atoi(getenv("TTUP_TIME_ONLY_OP")) : -1;      // debugging aid: time a single op
    for (int i = 0; i < n && rc == TTUP_OK; ++i) {
        if (only_op >= 0 && i != only_op) { ms_out[i] = 0.f; op_info(net, i, info_out + 8 * i, nullptr); continue; }
        const Op& op = net->ops[i];
        auto once = [&]() { return (op.kind == Op::UPSUM_HEAD || op.head) ? run_head_op(net, batch, net->heat_scratch, net->argmax_scratch, net->win_scratch, st) : run_op(net, op, batch, st); };
        rc = once();                       // warm-up launch of this op
        if (rc == TTUP_OK) {
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps && rc == TTUP_OK; ++r) rc = once();
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms_out[i] = ms / reps;
        }
        op_info(net, i, info_out + 8 * i, nullptr);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *n_ops_out = n;
    return rc;
}

// The whole graph in order, as the forward pass launches it (one micro-batch on lane 0, `stream`), with a HIP event between
// consecutive ops: ms_out[i] = average time from the end of op i-1 to the end of op i over `reps` passes, i.e. the launch
// duration of op i with the cache state it really sees.  This is what bench.py's `roofline` is computed from and what the
// rocprofv3 kernel trace of the same run (TTUP_LANES=1) reports per kernel.  names_out: max_ops x 64 chars.
extern "C" int ttup_wasb_time_graph(ttup_wasb* net, int batch, int reps, int max_ops, float* ms_out, int* info_out, char* names_out,
                                    int* n_ops_out, void* stream) {
    TTUP_REQUIRE(net && ms_out && info_out && n_ops_out, TTUP_EINVAL, "ttup_wasb_time_graph: null pointer");
    TTUP_REQUIRE(batch > 0 && batch <= net->micro && reps > 0, TTUP_EINVAL, "ttup_wasb_time_graph: batch must be in [1,%d]", net->micro);
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)net->ops.size();
    TTUP_REQUIRE(n <= max_ops, TTUP_EINVAL, "ttup_wasb_time_graph: %d ops exceed max_ops %d", n, max_ops);
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) TTUP_HIP_CHECK(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    std::vector<std::string> exact(n);          // the device kernel each op launched (kernel_note of its launcher), as rocprofv3 names it
    int rc = TTUP_OK;
    net->use_lane(0);
    for (int r = -1; r < reps && rc == TTUP_OK; ++r) {          // pass -1 = warm-up
        (void)hipEventRecord(ev[0], st);
        for (int i = 0; i < n && rc == TTUP_OK; ++i) {
            const Op& op = net->ops[i];
            kernel_note_reset();
            rc = (op.kind == Op::UPSUM_HEAD || op.head) ? run_head_op(net, batch, net->heat_scratch, net->argmax_scratch, net->win_scratch, st) : run_op(net, op, batch, st);
            (void)hipEventRecord(ev[i + 1], st);
            if (r < 0) exact[i] = kernel_noted();
        }
        (void)hipEventSynchronize(ev[n]);
        if (r >= 0) for (int i = 0; i < n; ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); acc[i] += ms; }
    }
    for (int i = 0; i < n; ++i) {
        ms_out[i] = (float)(acc[i] / reps);
        char nm[64];
        op_info(net, i, info_out + 8 * i, nm);
        if (names_out) {
            // "<device kernel template-id><+epilogue variant>": the op-level label keeps only its '+...' suffix when the launcher left a note
            std::string full = exact[i].empty() ? std::string(nm) : exact[i] + (strchr(nm, '+') ? strchr(nm, '+') : "");
            memset(names_out + 64 * i, 0, 64);
            memcpy(names_out + 64 * i, full.c_str(), full.size() < 63 ? full.size() : 63);
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *n_ops_out = n;
    return rc;
}

// The same launches back to back, `reps` passes between ONE pair of events (no event between the ops: an event record between two
// kernels is a packet of its own on the queue, and the per-op intervals of ttup_wasb_time_graph each include one).  ms_out[0] = the
// average duration of a pass: what one lane of the pipeline spends on a micro-batch.
extern "C" int ttup_wasb_time_replay(ttup_wasb* net, int batch, int reps, float* ms_out, void* stream) {
    TTUP_REQUIRE(net && ms_out, TTUP_EINVAL, "ttup_wasb_time_replay: null pointer");
    TTUP_REQUIRE(batch > 0 && batch <= net->micro && reps > 0, TTUP_EINVAL, "ttup_wasb_time_replay: batch must be in [1,%d]", net->micro);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    TTUP_HIP_CHECK(hipEventCreate(&e0));
    TTUP_HIP_CHECK(hipEventCreate(&e1));
    int rc = TTUP_OK;
    net->use_lane(0);
    for (int r = -1; r < reps && rc == TTUP_OK; ++r) {          // pass -1 = warm-up
        if (r == 0) (void)hipEventRecord(e0, st);
        for (const Op& op : net->ops) {
            rc = (op.kind == Op::UPSUM_HEAD || op.head) ? run_head_op(net, batch, net->heat_scratch, net->argmax_scratch, net->win_scratch, st) : run_op(net, op, batch, st);
            if (rc != TTUP_OK) break;
        }
    }
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    if (rc == TTUP_OK) { (void)hipEventElapsedTime(&ms, e0, e1); ms_out[0] = ms / reps; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

extern "C" int ttup_preprocess_triples(const uint8_t* frames_dev, int n_frames, int src_h, int src_w, int dst_h, int dst_w,
                                       float* out_dev, void* stream) {
    TTUP_REQUIRE(frames_dev && out_dev, TTUP_EINVAL, "ttup_preprocess_triples: null pointer");
    TTUP_REQUIRE(n_frames >= 3 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, TTUP_EINVAL, "ttup_preprocess_triples: bad shape");
    return launch_preprocess(frames_dev, n_frames, src_h, src_w, dst_h, dst_w, out_dev, TTUP_LAYOUT_NCHW_F32, TTUP_DTYPE_F32, 0, n_frames - 2, 3, (hipStream_t)stream);
}

extern "C" int ttup_preprocess_frames(const uint8_t* frames_dev, int n_frames, int src_h, int src_w, int dst_h, int dst_w,
                                      float* out_dev, void* stream) {
    TTUP_REQUIRE(frames_dev && out_dev, TTUP_EINVAL, "ttup_preprocess_frames: null pointer");
    TTUP_REQUIRE(n_frames >= 1 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, TTUP_EINVAL, "ttup_preprocess_frames: bad shape");
    return launch_preprocess(frames_dev, n_frames, src_h, src_w, dst_h, dst_w, out_dev, TTUP_LAYOUT_NCHW_F32, TTUP_DTYPE_F32, 0, n_frames, 1, (hipStream_t)stream);
}

// Scheduling priority of the handle's internal lane streams (high != 0: the greatest priority of the device).  Two handles that
// share the GPU (the hub path runs the table and the ball detector side by side) can thus be ordered: the high-priority one
// finishes first and its host-side consumer overlaps with the other's kernels.  Synchronises; single-lane handles run on the
// caller's stream and are not affected.
extern "C" int ttup_wasb_set_priority(ttup_wasb* net, int high) {
    TTUP_REQUIRE(net, TTUP_EINVAL, "ttup_wasb_set_priority: null handle");
    if (net->lanes.size() < 2) return TTUP_OK;
    TTUP_HIP_CHECK(hipDeviceSynchronize());
    int least = 0, greatest = 0;
    TTUP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (auto& L : net->lanes) {
        if (L.stream) (void)hipStreamDestroy(L.stream);
        L.stream = nullptr;
        TTUP_HIP_CHECK(hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, high ? greatest : least));
    }
    return TTUP_OK;
}

extern "C" int ttup_wasb_micro_batch(ttup_wasb* net) { return net ? net->micro : 0; }
extern "C" int ttup_wasb_out_channels(ttup_wasb* net) { return net ? net->n_out : 0; }
