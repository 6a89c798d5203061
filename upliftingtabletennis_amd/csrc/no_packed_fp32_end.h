// closes no_packed_fp32_begin.h
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
