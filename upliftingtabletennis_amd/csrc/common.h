// Shared helpers of libttup.so (error reporting, HIP checks, bf16 bit tricks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string>

#include "../../include/ttup.h"

namespace ttup {

void set_error(const char* fmt, ...);   // thread-local, returned by ttup_last_error()
// Measurement aid (ttup_wasb_time_graph / time_ops -> bench.py's roofline and its lookup of PMC traffic by kernel): a launcher
// leaves the template-id of the device kernel it launches exactly as rocprofv3 prints it ("conv_mfma_kernel<32, 128, 3, 1, 8, 32, 8, false>").
// Thread-local; only the FIRST note after a reset is kept (an op's main kernel comes first, a tiny finishing kernel may follow).
void kernel_note(const char* fmt, ...);
void kernel_note_reset();
const char* kernel_noted();

#define TTUP_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ttup::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TTUP_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

#define TTUP_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            ttup::set_error(__VA_ARGS__);                                                     \
            return code;                                                                      \
        }                                                                                     \
    } while (0)

#define TTUP_LAUNCH_CHECK()  TTUP_HIP_CHECK(hipGetLastError())

// Packed fp32 instructions WITH operand swizzles (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 carrying op_sel, op_sel_hi or neg
// modifiers) return wrong values on MI355X (gfx950, ROCm 7.2.0) while a wave of ANOTHER kernel on the same CU feeds MFMAs from LDS
// reads -- which the CNN's chain kernels do all the time.  Measured with tools/pk_coresidency_repro.hip: 24 % of the results of a
// 20-line rotation kernel are wrong beside a 20-line ds_read -> MFMA loop, none with the same source compiled without packed fp32;
// the plain element-wise forms (all that the convolution epilogues contain) are not affected.  Translation units whose fp32
// vector code the compiler turns into swizzled packed forms (the uplift transformer's RoPE / softmax arithmetic, the refine fit)
// include no_packed_fp32_begin.h before anything else and no_packed_fp32_end.h last (whole unit), or bracket a few self-contained
// kernels with the two macros below -- those must not call HIP header functions (a callee without the attribute is not inlined:
// no_packed_fp32_begin.h); tests/test_cabi.py checks the device ISA of the whole library for swizzled packed fp32 and for calls.
#if defined(__HIP_DEVICE_COMPILE__)
#define TTUP_NO_PACKED_FP32_BEGIN _Pragma("clang attribute push(__attribute__((target(\"no-packed-fp32-ops\"))), apply_to = function)")
#define TTUP_NO_PACKED_FP32_END _Pragma("clang attribute pop")
#else
#define TTUP_NO_PACKED_FP32_BEGIN
#define TTUP_NO_PACKED_FP32_END
#endif

// threadIdx.x & co. as builtins: the HIP accessors go through __ockl_get_local_id / __ockl_get_group_id of the device library,
// which a unit compiled with no-packed-fp32-ops cannot inline (no_packed_fp32_begin.h)
#if defined(__HIP_DEVICE_COMPILE__)
#define TTUP_DEV_BUILTIN(x) (int)(x)
#else
#define TTUP_DEV_BUILTIN(x) 0          // (host pass: the bodies are parsed, never run)
#endif
__device__ __forceinline__ int ttup_tid_x() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_workitem_id_x()); }
__device__ __forceinline__ int ttup_bid_x() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_workgroup_id_x()); }
__device__ __forceinline__ int ttup_bid_y() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_workgroup_id_y()); }
__device__ __forceinline__ int ttup_bid_z() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_workgroup_id_z()); }
__device__ __forceinline__ int ttup_bdim_x() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_workgroup_size_x()); }
__device__ __forceinline__ int ttup_gsize_x() { return TTUP_DEV_BUILTIN(__builtin_amdgcn_grid_size_x()); }          // gridDim.x * blockDim.x

typedef uint16_t bf16_t;   // raw bits

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)v << 16;
    return x.f;
}
// round-to-nearest-even, NaN preserved
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } x;
    x.f = f;
    if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);
    return (bf16_t)((x.u + 0x7fffu + ((x.u >> 16) & 1u)) >> 16);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Per-device one-time setup (api.hip).  Kernels that need more than 64 KB of dynamic LDS must have
// hipFuncAttributeMaxDynamicSharedMemorySize raised on EVERY device they are launched on, and handles may live on any
// device of the process: both helpers key their state by hipGetDevice() under a mutex (no process-global flags).
int ensure_max_lds(const void* kernel, size_t bytes);          // TTUP_OK or TTUP_EHIP
int device_normalise_lut(const float** lut_dev);               // [3][256] (v/255 - mean[c]) / std[c] on the current device

}  // namespace ttup
