// Library-level entry points: version, error string, device probe.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace ttup {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static thread_local char g_kernel[64] = "";
void kernel_note(const char* fmt, ...) {
    if (g_kernel[0]) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof g_kernel, fmt, ap);
    va_end(ap);
}
void kernel_note_reset() { g_kernel[0] = 0; }
const char* kernel_noted() { return g_kernel; }

int ensure_max_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return TTUP_OK;
    int dev = 0;
    TTUP_HIP_CHECK(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find({kernel, dev});
    if (it != done.end() && it->second >= bytes) return TTUP_OK;
    TTUP_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[{kernel, dev}] = bytes;
    return TTUP_OK;
}

int device_normalise_lut(const float** lut_dev) {
    int dev = 0;
    TTUP_HIP_CHECK(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<int, float*> luts;
    std::lock_guard<std::mutex> lk(mu);
    auto it = luts.find(dev);
    if (it == luts.end()) {
        // ImageNet mean / std of balldetection/transforms.py:388-401, evaluated in fp64 like the reference, rounded to fp32
        const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
        float h[3 * 256];
        for (int c = 0; c < 3; ++c) for (int v = 0; v < 256; ++v) h[c * 256 + v] = (float)(((double)v / 255.0 - mean[c]) / sd[c]);
        float* d = nullptr;
        TTUP_HIP_CHECK(hipMalloc((void**)&d, sizeof h));
        TTUP_HIP_CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
        it = luts.emplace(dev, d).first;
    }
    *lut_dev = it->second;
    return TTUP_OK;
}
}  // namespace ttup

#ifndef TTUP_BUILD_ID
#error "TTUP_BUILD_ID is not defined: build with `python -m upliftingtabletennis_amd.build` (it hashes the sources into the library)"
#endif
extern "C" int ttup_version(void) { return 103; }      // 102 (round 6): ttup_wasb_certify_audit_crops, certify_stats copies twelve counters, up to 32 crops per heatmap; 103: ttup_wasb_time_replay
extern "C" const char* ttup_build_id(void) { return TTUP_BUILD_ID; }
extern "C" const char* ttup_last_error(void) { return ttup::get_error(); }
extern "C" int ttup_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { ttup::set_error("hipGetDeviceCount failed"); return -1; }
    return n;
}
