"""Measured peaks of the device the path runs on (csrc/peaks.hip; SURVEY 8d "Peaks to divide by"): what a register-resident bf16
MFMA loop and streaming HBM kernels reach on THIS GPU, reported beside the vendor figures the roofline fractions are quoted
against (2.5 PFLOP/s dense bf16, 8 TB/s).  No reference counterpart; measurement only."""
import ctypes

import numpy as np
import torch

from . import _lib


def mfma_bf16(iters=20000, waves_per_simd=2, device='cuda:0'):
    """-> {'tflops_16x16x32', 'tflops_32x32x16', 'ms_16x16x32', 'ms_32x32x16', 'waves_per_simd'} on random operands."""
    _lib.require_gpu()
    out = np.zeros(4, np.float64)
    with torch.cuda.device(torch.device(device)):
        _lib.check(_lib.load().ttup_peak_mfma_bf16(int(iters), int(waves_per_simd), out.ctypes.data_as(ctypes.c_void_p), _lib.stream_ptr()))
    return {'tflops_16x16x32': float(out[0]), 'tflops_32x32x16': float(out[1]), 'ms_16x16x32': float(out[2]), 'ms_32x32x16': float(out[3]),
            'waves_per_simd': waves_per_simd, 'operands': 'random bf16 in registers, 16 / 4 independent accumulator chains per wave'}


def hbm(bytes_per_array=1 << 30, device='cuda:0'):
    """-> GB/s of a streaming read, copy and triad over arrays of `bytes_per_array` (default 1 GiB each: past the Infinity Cache)."""
    _lib.require_gpu()
    out = np.zeros(6, np.float64)
    with torch.cuda.device(torch.device(device)):
        _lib.check(_lib.load().ttup_peak_hbm(int(bytes_per_array), out.ctypes.data_as(ctypes.c_void_p), _lib.stream_ptr()))
    return {'read_gbs': float(out[0]), 'copy_gbs': float(out[1]), 'triad_gbs': float(out[2]), 'read_ms': float(out[3]), 'copy_ms': float(out[4]),
            'triad_ms': float(out[5]), 'bytes_per_array': int(bytes_per_array), 'access': '16 bytes per lane, grid-stride, 4096 workgroups of 256'}


def measure(device='cuda:0'):
    m = mfma_bf16(device=device)
    m1 = mfma_bf16(waves_per_simd=1, device=device)
    h = hbm(device=device)
    return {'mfma_bf16': m, 'mfma_bf16_one_wave_per_simd': m1, 'hbm': h,
            'peak_bf16_tflops': max(m['tflops_16x16x32'], m['tflops_32x32x16'], m1['tflops_16x16x32'], m1['tflops_32x32x16']),
            'peak_hbm_gbs': max(h['read_gbs'], h['copy_gbs'], h['triad_gbs']),
            'vendor': {'bf16_tflops_dense': 2500.0, 'hbm_gbs': 8000.0}}
