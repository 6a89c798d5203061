// Stand-alone reproducer of the co-residency finding behind csrc/common.h's TTUP_NO_PACKED_FP32_* (DESIGN.md 12).
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math -o pk_coresidency_repro tools/pk_coresidency_repro.hip && ./pk_coresidency_repro
// Victim: a 2-D rotation of random operands (the RoPE arithmetic of the uplift's attention kernel), results folded into one word per
// thread -- once as the compiler emits it by default (v_pk_mul_f32 / v_pk_fma_f32 with op_sel / op_sel_hi / neg modifiers), once
// with packed fp32 instructions disabled for the function, and an element-wise kernel whose packed instructions carry NO operand
// swizzles.  Neighbours on another stream: MFMAs fed from LDS reads, a register-resident MFMA loop, LDS reads into integer ALU ops,
// a plain VALU loop.  Every victim result is compared with the result of the same kernel on an idle GPU.
// Streams map onto a few hardware queues round-robin; a stream that shares the neighbour's queue is serialised behind it and
// shows nothing, so four victim streams are tried and each is reported.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(unsigned* p, size_t n) {       // bf16 pairs in [-2, 2) = fp32 values of moderate size
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ 0x9e3779b9u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned lo = (x & 0x807fu) | ((125u + (x >> 8) % 3u) << 7), hi = ((x >> 16) & 0x807fu) | ((125u + (x >> 24) % 3u) << 7);
        p[i] = lo | (hi << 16);
    }
}

// ---------------------------------------------------------------- neighbours
__global__ __launch_bounds__(256) void lds_mfma_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, int lds_units) {
    extern __shared__ __attribute__((aligned(16))) uint4 sm4[];
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    for (int u = threadIdx.x; u < lds_units; u += 256) sm4[u] = seed[(gw * 64 + u) % (4096 * 64)];
    __syncthreads();
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 a = __builtin_bit_cast(bf16x8, seed[gw % 4096 * 64 + lane]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, sm4[(threadIdx.x + (it * 8 + k) * 67) % lds_units]);
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        }
    }
    f32x4 s = acc[0];
    for (int k = 1; k < 8; ++k) s += acc[k];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[gw] = s[0];
}
__global__ __launch_bounds__(256) void reg_mfma_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, int) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    bf16x8 a[4], b[4];
    for (int k = 0; k < 4; ++k) {
        a[k] = __builtin_bit_cast(bf16x8, seed[(gw * 8 + k) % 4096 * 64 + lane]);
        b[k] = __builtin_bit_cast(bf16x8, seed[(gw * 8 + 4 + k) % 4096 * 64 + lane]);
    }
    f32x4 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b[k >> 2], acc[k], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int k = 1; k < 16; ++k) s += acc[k];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[gw] = s[0];
}
__global__ __launch_bounds__(256) void lds_xor_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, int lds_units) {
    extern __shared__ __attribute__((aligned(16))) uint4 sm4[];
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    for (int u = threadIdx.x; u < lds_units; u += 256) sm4[u] = seed[(gw * 64 + u) % (4096 * 64)];
    __syncthreads();
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 b = sm4[(threadIdx.x + (it * 8 + k) * 67) % lds_units];
            acc[k] += b.x ^ b.y ^ b.z ^ b.w;
        }
    }
    unsigned s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k];
    if (s == 0x12345678u) sink[gw] = 1.f;
}
__global__ __launch_bounds__(256) void valu_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, int) {
    const int gw = blockIdx.x * 256 + threadIdx.x;
    float x[16];
    for (int k = 0; k < 16; ++k) x[k] = __uint_as_float((seed[(gw + k) % 4096].x & 0x007fffffu) | 0x3f000000u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = fmaf(x[k], 0.999f, 0.0005f);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += x[k];
    if (s == 12345.678f) sink[gw >> 6] = s;
}

// ---------------------------------------------------------------- victims
#define ROT_BODY                                                                                                                     \
    const int t = blockIdx.x * 64 + threadIdx.x;                                                                                     \
    unsigned h = 0;                                                                                                                  \
    for (int r = 0; r < reps; ++r) {                                                                                                 \
        const int i = (t + r * 4099) % n;                                                                                            \
        const f32x4 k = a[i], cs = b[i];                                                                                             \
        const f32x4 o = f32x4{k[0] * cs[0] - k[1] * cs[1], k[0] * cs[1] + k[1] * cs[0], k[2] * cs[2] - k[3] * cs[3], k[2] * cs[3] + k[3] * cs[2]}; \
        h = h * 31u + (__float_as_uint(o[0]) ^ __float_as_uint(o[1]) * 3u ^ __float_as_uint(o[2]) * 5u ^ __float_as_uint(o[3]) * 7u); \
    }                                                                                                                                \
    out[t] = h;
__global__ __launch_bounds__(64) void rot_packed_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, unsigned* __restrict__ out, int n, int reps) { ROT_BODY }
__global__ __launch_bounds__(64) void ew_packed_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, unsigned* __restrict__ out, int n, int reps) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    unsigned h = 0;
    for (int r = 0; r < reps; ++r) {
        const int i = (t + r * 4099) % n;
        const f32x4 k = a[i], cs = b[i];
        f32x4 o = k * cs + k;
        o = o * o + cs;
        o = o + k;
        h = h * 31u + (__float_as_uint(o[0]) ^ __float_as_uint(o[1]) * 3u ^ __float_as_uint(o[2]) * 5u ^ __float_as_uint(o[3]) * 7u);
    }
    out[t] = h;
}
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif
__global__ __launch_bounds__(64) void rot_plain_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, unsigned* __restrict__ out, int n, int reps) { ROT_BODY }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif

int main() {
    const size_t seed_words = (size_t)4096 * 64 * 4;
    uint4* seed; float* sink; unsigned* out;
    const int BLOCKS = 2048, REPS = 400, N = 4096 * 32, NOUT = BLOCKS * 64, LDS = 40 * 1024;
    CHECK(hipMalloc((void**)&seed, seed_words * 4));
    CHECK(hipMalloc((void**)&sink, 65536 * 16));
    CHECK(hipMalloc((void**)&out, NOUT * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)seed, seed_words);
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    hipStream_t nb_stream, vs[4];
    CHECK(hipStreamCreate(&nb_stream));
    for (auto& s : vs) CHECK(hipStreamCreate(&s));
    typedef void (*nb_fn)(const uint4*, float*, int, int);
    struct { const char* name; nb_fn fn; int iters, blocks, lds; } nbs[4] = {
        {"LDS reads -> MFMA", lds_mfma_kernel, 800, 768, LDS}, {"register-resident MFMA loop", reg_mfma_kernel, 430, 1024, 0},
        {"LDS reads -> integer XOR", lds_xor_kernel, 800, 768, LDS}, {"VALU loop", valu_kernel, 215, 1024, 0}};
    typedef void (*v_fn)(const f32x4*, const f32x4*, unsigned*, int, int);
    struct { const char* name; v_fn fn; } victims[3] = {{"rotation, packed fp32 with operand swizzles", rot_packed_kernel},
                                                       {"rotation, packed fp32 disabled", rot_plain_kernel},
                                                       {"element-wise, packed fp32 without swizzles", ew_packed_kernel}};
    const f32x4* a = (const f32x4*)seed;
    std::vector<unsigned> ref(NOUT), got(NOUT);
    int wrong_total[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v) {
        hipLaunchKernelGGL(victims[v].fn, dim3(BLOCKS), dim3(64), 0, vs[0], a, a + N, out, N, REPS);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ref.data(), out, NOUT * 4, hipMemcpyDeviceToHost));
        for (int nb = 0; nb < 4; ++nb) {
            for (int si = 0; si < 4; ++si) {
                long long words = 0; int runs = 0;
                for (int rep = 0; rep < 10; ++rep) {
                    for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(nbs[nb].fn, dim3(nbs[nb].blocks), dim3(256), nbs[nb].lds, nb_stream, seed, sink, nbs[nb].iters, nbs[nb].lds / 16);
                    hipLaunchKernelGGL(victims[v].fn, dim3(BLOCKS), dim3(64), 0, vs[si], a, a + N, out, N, REPS);
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(got.data(), out, NOUT * 4, hipMemcpyDeviceToHost));
                    long long d = 0;
                    for (int i = 0; i < NOUT; ++i) d += got[i] != ref[i];
                    words += d; runs += d > 0;
                }
                printf("victim %-44s neighbour %-28s stream %d: %2d of 10 runs differ, %lld of %d words wrong\n", victims[v].name, nbs[nb].name, si, runs, words, 10 * NOUT);
                wrong_total[v] += runs;
            }
        }
    }
    printf("summary: runs with wrong results -- swizzled packed fp32: %d, packed fp32 disabled: %d, packed fp32 without swizzles: %d (of 160 each)\n",
           wrong_total[0], wrong_total[1], wrong_total[2]);
    return 0;
}
