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
#include "common.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <memory>
#include <vector>

using namespace ttup;

TTUP_NO_PACKED_FP32_BEGIN      // this unit's kernels run beside the CNN's chain kernels (common.h)

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------ packed linear layer
struct Linear {
    int n = 0, k = 0;            // out features, in features
    float* w_dev = nullptr;      // MFMA path: [ntile][k/16][64 lanes][4]; small-K path: [n][k] row major
    float* b_dev = nullptr;      // [n] or null
    bool mfma = false;
};

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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * (64 * NTW);
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
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
                const float mean = sum / (float)K;
                float var = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (4 * (l16 + 16 * u) >= K) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; var = fmaf(d, d, var); }
                }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) var += __shfl_xor(var, off, 64);
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

// out[m][n] = relu?(sum_k x[m][k] w[n][k] + b[n]) for tiny K (2 or 3): embedding fc1
__global__ void small_linear_kernel(const float* x, int ldx, const float* w, const float* b, float* out, int ldo, long long M, int N, int K, int relu) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
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
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
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

// P threads per (sequence, head); a workgroup of blockDim.x threads serves blockDim.x / P sequences.  K (rotated) and V
// of each sequence live in LDS, thread i0 owns query rows i0, i0+P, ...
template <int HD, int P>
__global__ __launch_bounds__(128) void attention_kernel(AttnArgs a) {      // at most 128 threads are ever launched: 256 VGPRs, no spills
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = a.S, G = blockDim.x / P;
    const int SEQ = 2 * S * HD + 16;                 // floats per sequence; the +16 words spreads the groups over LDS banks
    float* ms = sm + G * SEQ;                        // [G][S] additive mask
    const int h = blockIdx.y, tid = threadIdx.x;
    const int D3 = 3 * a.D, HV = HD / 4;
    // ---- stage K (RoPE applied) and V, one float4 per thread per step, 128 B rows read by HV consecutive threads
    for (int u = tid; u < G * S * HV; u += blockDim.x) {
        const int g = u / (S * HV), rem = u - g * (S * HV), j = rem / HV, part = rem - j * HV;
        const int seq = blockIdx.x * G + g;
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
    for (int u = tid; u < G * S; u += blockDim.x) {
        const int seq = blockIdx.x * G + u / S;
        ms[u] = seq < a.n_seq ? a.mask[(size_t)(seq / a.mask_div) * S + (u % S)] : -INFINITY;
    }
    __syncthreads();
    const int g = tid / P, i0 = tid - g * P;
    const int seq = blockIdx.x * G + g;
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

// ------------------------------------------------------------------ token assembly helpers
// x[(b,t), 0] = ball_tok[b,t]; x[(b,t), 1+n] = table_tok[b,n]      (model.py:374-378)
__global__ void assemble_table_kernel(const float* ball_tok, const float* table_tok, float* x, int T, int NT, int D, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    long long r = i / D;
    const int n = (int)(r % (NT + 1)); r /= (NT + 1);      // r = b*T + t
    x[i] = n == 0 ? ball_tok[r * D + d] : table_tok[((r / T) * NT + (n - 1)) * D + d];
}
// y[r] = x[r*stride_tok] rows (token 0 of every sequence)            (model.py:383-384)
__global__ void gather_rows_kernel(const float* x, float* y, int D, int seq_tokens, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const long long r = i / D;
    y[i] = x[(r * seq_tokens) * D + d];
}
// y[b, 0] = cls; y[b, 1+t] = x[b, t]                                 (model.py:560)
__global__ void prepend_cls_kernel(const float* x, const float* cls, float* y, int T, int D, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    long long r = i / D;
    const int t = (int)(r % (T + 1)); const long long b = r / (T + 1);
    y[i] = t == 0 ? cls[d] : x[(b * T + (t - 1)) * D + d];
}
// masks: mask (B,T) {0,1} -> additive m1 (B,T), m2 (B,T+1) with leading 0; table (B,13,3) -> tmask (B,14), txy (B*13,2)
__global__ void prepare_kernel(const float* mask, const float* table, float* m1, float* m2, float* tmask, float* txy, int B, int T, int NT, int* flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nmask = (long long)B * T, ntab = (long long)B * NT;
    if (i < nmask) {
        const float m = mask[i];
        const float add = m == 0.f ? -INFINITY : 0.f;
        m1[i] = add;
        const long long b = i / T; const int t = (int)(i % T);
        m2[b * (T + 1) + 1 + t] = add;
        if (t == 0) m2[b * (T + 1)] = 0.f;
        // bit0: some m==0, bit1: some m==1, bit2: some m<0, bit3: some m>1  (min==0 && max==1  <=>  flags==3)
        if (m == 0.f) atomicOr(flags, 1);
        else if (m == 1.f) atomicOr(flags, 2);
        else if (m < 0.f) atomicOr(flags, 4);
        else if (m > 1.f || m != m) atomicOr(flags, 8);
    } else if (i < nmask + ntab) {
        const long long j = i - nmask;
        const long long b = j / NT; const int n = (int)(j % NT);
        tmask[b * (NT + 1) + 1 + n] = table[j * 3 + 2] == 1.f ? 0.f : -INFINITY;      // KEYPOINT_VISIBLE == 1, model.py:363
        if (n == 0) tmask[b * (NT + 1)] = 0.f;
        txy[j * 2] = table[j * 3]; txy[j * 2 + 1] = table[j * 3 + 1];
    }
}
__global__ void rotationaxes_kernel(const float* rot, const float* pos, int B, int T, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
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
    float2 *rope = nullptr, *table_rope = nullptr;      // (cos, sin) tables: [chunk*max_len][hd/2] per forward, [n_table][hd/2] fixed
    std::vector<void*> allocs;
    // scratch (sized for `chunk` trajectories of max_len tokens)
    float *x = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *x2 = nullptr, *tok = nullptr, *ttok = nullptr, *h1 = nullptr;
    float *m1 = nullptr, *m2 = nullptr, *tmask = nullptr, *txy = nullptr, *tmp_small = nullptr;
    int* flags_dev = nullptr;
    ~ttup_uplift() { for (void* p : allocs) if (p) (void)hipFree(p); }
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
    const size_t smem = (size_t)4 * bm * (L.k / 4 + 4) * sizeof(float);
    const dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((L.n + 64 * ntw - 1) / (64 * ntw)));
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
    if ((rc = run_linear(L.qkv, x, D, tokens, L.g1, L.b1, 0, nullptr, 0, net->qkv, 3 * D, st))) return rc;
    if ((rc = run_attention(net, net->qkv, net->att, n_seq, S, num_cls, mask, mask_div, rope, times_div, times_stride, st))) return rc;
    if ((rc = run_linear(L.proj, net->att, D, tokens, nullptr, nullptr, 0, x, D, net->x2, D, st))) return rc;       // x2 = proj(att) + x
    if ((rc = run_linear(L.fc1, net->x2, D, tokens, L.g2, L.b2, 1, nullptr, 0, net->hid, D, st))) return rc;          // hid = relu(fc1(LN(x2)))
    return run_linear(L.fc2, net->hid, D, tokens, nullptr, nullptr, 0, net->x2, D, x, D, st);                       // x = fc2(hid) + x2
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
    // table stage
    const long long tok1 = (long long)B * T * S1;
    {
        const long long total = tok1 * D;
        hipLaunchKernelGGL(assemble_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, net->tok, net->ttok, net->x, T, NT, D, total);
        TTUP_LAUNCH_CHECK();
    }
    for (const Layer& L : net->pos_layers)
        if ((rc = run_layer(net, L, net->x, tok1, B * T, S1, 1, net->tmask, T, net->table_rope, 1, 0, st))) return rc;
    {
        const long long total = (long long)B * T * D;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, net->x, net->tok, D, S1, total);
        TTUP_LAUNCH_CHECK();
    }
    // temporal stage (tok is [B*T][D])
    for (const Layer& L : net->layers)
        if ((rc = run_layer(net, L, net->tok, (long long)B * T, B, T, 0, net->m1, 1, net->rope, 1, T, st))) return rc;
    if ((rc = run_head(net, net->position_head, net->tok, D, (long long)B * T, pos, st))) return rc;
    // spin stage
    {
        const long long total = (long long)B * (T + 1) * D;
        hipLaunchKernelGGL(prepend_cls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, net->tok, net->cls_dev, net->x, T, D, total);
        TTUP_LAUNCH_CHECK();
    }
    for (const Layer& L : net->second)
        if ((rc = run_layer(net, L, net->x, (long long)B * (T + 1), B, T + 1, 1, net->m2, 1, net->rope, 1, T, st))) return rc;
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
    for (int b0 = 0; b0 < batch; b0 += (int)cap) {
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

extern "C" int ttup_transform_rotationaxes(const float* rot_dev, const float* pos_dev, int batch, int len, float* out_dev, void* stream) {
    TTUP_REQUIRE(rot_dev && pos_dev && out_dev, TTUP_EINVAL, "ttup_transform_rotationaxes: null pointer");
    TTUP_REQUIRE(batch >= 0 && len >= 2, TTUP_EINVAL, "ttup_transform_rotationaxes: need at least two positions");
    if (batch == 0) return TTUP_OK;
    hipLaunchKernelGGL(rotationaxes_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, rot_dev, pos_dev, batch, len, out_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

TTUP_NO_PACKED_FP32_END
