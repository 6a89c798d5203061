// Measured peaks of the device the path runs on (SURVEY 8d "Peaks to divide by"): a register-resident bf16 MFMA loop and
// streaming HBM kernels, timed with HIP events on the caller's stream.  Measurement aids, not part of the hot path; the reference
// has no counterpart.  Random operands on purpose: the chip holds a lower clock on random data than on zeros
// (MI355X_MICROARCH.md, DVFS give-back), and that clock is the one the convolutions get.
#include "common.h"

namespace ttup {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// 16 independent accumulator chains of v_mfma_f32_16x16x32_bf16 per wave, operands in registers
__global__ __launch_bounds__(256) void mfma16_loop_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = __builtin_bit_cast(bf16x8, seed[(gw * 8 + k) % 4096 * 64 + lane]);
        b[k] = __builtin_bit_cast(bf16x8, seed[(gw * 8 + 4 + k) % 4096 * 64 + lane]);
    }
    f32x4 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b[k >> 2], acc[k], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += acc[k];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[gw] = s[0];        // keeps the loop alive, (almost) never stores
}

// 4 independent chains of v_mfma_f32_32x32x16_bf16 per wave
__global__ __launch_bounds__(256) void mfma32_loop_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    bf16x8 a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        a[k] = __builtin_bit_cast(bf16x8, seed[(gw * 4 + k) % 4096 * 64 + lane]);
        b[k] = __builtin_bit_cast(bf16x8, seed[(gw * 4 + 2 + k) % 4096 * 64 + lane]);
    }
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 1], b[k >> 1], acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[k][j];
    if (s == 12345.678f) sink[gw] = s;
}

// streaming kernels, 16 bytes per lane, grid-stride
__global__ __launch_bounds__(256) void hbm_read_kernel(const float4* __restrict__ a, float* __restrict__ sink, size_t n4) {
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}
__global__ __launch_bounds__(256) void hbm_copy_kernel(const float4* __restrict__ a, float4* __restrict__ c, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) c[i] = a[i];
}
__global__ __launch_bounds__(256) void hbm_triad_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, float s, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        c[i] = float4{x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w};
    }
}
__global__ void fill_random_kernel(unsigned* p, size_t n, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ salt;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        // two bf16 values in [-2, 2): sign + exponent 127 / 126 / 125 + random mantissa
        const unsigned lo = (x & 0x807fu) | ((125u + (x >> 8) % 3u) << 7), hi = ((x >> 16) & 0x807fu) | ((125u + (x >> 24) % 3u) << 7);
        p[i] = lo | (hi << 16);
    }
}

template <typename F>
int timed(hipStream_t st, int reps, double* ms_out, F launch) {
    hipEvent_t e0, e1;
    TTUP_HIP_CHECK(hipEventCreate(&e0));
    TTUP_HIP_CHECK(hipEventCreate(&e1));
    launch();                                       // warm-up (and clock ramp)
    launch();
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, st);
    const hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) { set_error("peak measurement failed: %s", hipGetErrorString(e)); return TTUP_EHIP; }
    TTUP_LAUNCH_CHECK();
    *ms_out = ms / reps;
    return TTUP_OK;
}

}  // namespace
}  // namespace ttup

using namespace ttup;

// out_host[0..1]: TFLOP/s of the v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 loops (8 waves per CU-SIMD pair:
// `waves_per_simd` waves on every SIMD of every CU); out_host[2..3]: their durations in ms.
extern "C" int ttup_peak_mfma_bf16(int iters, int waves_per_simd, double* out_host, void* stream) {
    TTUP_REQUIRE(out_host && iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 8, TTUP_EINVAL, "ttup_peak_mfma_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t prop;
    TTUP_HIP_CHECK(hipGetDevice(&dev));
    TTUP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount * waves_per_simd;          // 256 threads = one wave per SIMD
    uint4* seed = nullptr; float* sink = nullptr;
    const size_t seed_words = (size_t)4096 * 64 * 4;
    TTUP_HIP_CHECK(hipMalloc((void**)&seed, seed_words * 4));
    TTUP_HIP_CHECK(hipMalloc((void**)&sink, (size_t)blocks * 4 * sizeof(float)));
    hipLaunchKernelGGL(fill_random_kernel, dim3(1024), dim3(256), 0, st, (unsigned*)seed, seed_words, 0x9e3779b9u);
    double ms16 = 0, ms32 = 0;
    int rc = timed(st, 3, &ms16, [&]() { hipLaunchKernelGGL(mfma16_loop_kernel, dim3(blocks), dim3(256), 0, st, seed, sink, iters); });
    if (rc == TTUP_OK) rc = timed(st, 3, &ms32, [&]() { hipLaunchKernelGGL(mfma32_loop_kernel, dim3(blocks), dim3(256), 0, st, seed, sink, iters); });
    (void)hipFree(seed); (void)hipFree(sink);
    if (rc) return rc;
    const double waves = (double)blocks * 4;
    out_host[0] = waves * iters * 16.0 * (2.0 * 16 * 16 * 32) / (ms16 * 1e-3) / 1e12;
    out_host[1] = waves * iters * 8.0 * (2.0 * 32 * 32 * 16) / (ms32 * 1e-3) / 1e12;
    out_host[2] = ms16; out_host[3] = ms32;
    return TTUP_OK;
}

// out_host[0..2]: GB/s of a streaming read (bytes), copy (2 x bytes) and triad (3 x bytes) over arrays of `bytes` each
// (use >= 1 GiB: far past the 256 MiB Infinity Cache); out_host[3..5]: their durations in ms.
extern "C" int ttup_peak_hbm(size_t bytes, double* out_host, void* stream) {
    TTUP_REQUIRE(out_host && bytes >= (1u << 20) && bytes % 16 == 0, TTUP_EINVAL, "ttup_peak_hbm: bytes must be a multiple of 16, at least 1 MiB");
    hipStream_t st = (hipStream_t)stream;
    float4 *a = nullptr, *b = nullptr, *c = nullptr; float* sink = nullptr;
    TTUP_HIP_CHECK(hipMalloc((void**)&a, bytes));
    TTUP_HIP_CHECK(hipMalloc((void**)&b, bytes));
    TTUP_HIP_CHECK(hipMalloc((void**)&c, bytes));
    TTUP_HIP_CHECK(hipMalloc((void**)&sink, 64));
    hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, st, (unsigned*)a, bytes / 4, 1u);
    hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, st, (unsigned*)b, bytes / 4, 2u);
    const size_t n4 = bytes / 16;
    const int grid = 256 * 16;
    double ms[3] = {0, 0, 0};
    int rc = timed(st, 5, &ms[0], [&]() { hipLaunchKernelGGL(hbm_read_kernel, dim3(grid), dim3(256), 0, st, a, sink, n4); });
    if (rc == TTUP_OK) rc = timed(st, 5, &ms[1], [&]() { hipLaunchKernelGGL(hbm_copy_kernel, dim3(grid), dim3(256), 0, st, a, c, n4); });
    if (rc == TTUP_OK) rc = timed(st, 5, &ms[2], [&]() { hipLaunchKernelGGL(hbm_triad_kernel, dim3(grid), dim3(256), 0, st, a, b, c, 1.5f, n4); });
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(sink);
    if (rc) return rc;
    for (int k = 0; k < 3; ++k) { out_host[k] = (double)bytes * (k + 1) / (ms[k] * 1e-3) / 1e9; out_host[3 + k] = ms[k]; }
    return TTUP_OK;
}
