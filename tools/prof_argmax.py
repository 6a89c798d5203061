#!/usr/bin/env python3
"""Heatmap argmax / window kernel alone, for rocprofv3 PMC passes (FETCH_SIZE of argmax_partial_kernel): 256 fp32 heatmaps of
704x1280 (923 MB, past the 256 MiB Infinity Cache), a few repeats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import refine, _lib
heat = torch.randn((256, 704, 1280), device='cuda')
for _ in range(3):
    refine.refine_device(heat, 1920, 1080, _lib.REFINE_BALL)
torch.cuda.synchronize()
print('done')
