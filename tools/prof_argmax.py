#!/usr/bin/env python3
"""The kernels that stream fp32 heatmaps, alone, for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE): 256 fp32 heatmaps of 704x1280
(923 MB, past the 256 MiB Infinity Cache), a few repeats of (a) the standalone argmax + window seam (argmax_partial_kernel) and
(b) the certified argmax's candidate scan, the heatmap pass of the timed path (cert_scan_kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import refine, _lib
n, h, w, K = 256, 704, 1280, 32
heat = torch.randn((n, h, w), device='cuda')
for _ in range(3):
    _, idx, _ = refine.refine_device(heat, 1920, 1080, _lib.REFINE_BALL)
lib = _lib.load()
cidx = torch.empty((n, K), dtype=torch.int32, device='cuda')
cbf = torch.empty((n, K), dtype=torch.float32, device='cuda')
for _ in range(3):
    ccnt = torch.zeros((n,), dtype=torch.int32, device='cuda')
    _lib.check(lib.ttup_certify_scan(_lib.ptr(heat), _lib.ptr(idx), n, h, w, 0.05, K, _lib.ptr(cidx), _lib.ptr(ccnt), _lib.ptr(cbf), _lib.stream_ptr()))
torch.cuda.synchronize()
print('done')
