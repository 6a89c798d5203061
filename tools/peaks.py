#!/usr/bin/env python3
"""Measured device peaks -> profiles/<tag>_peaks.json (tools/peaks.py [tag]; default tag r3).  Run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import peaks  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'r3'
res = peaks.measure()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('gpurun_out', 'profiles'):
    os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, d, '%s_peaks.json' % tag), 'w') as f:
        json.dump(res, f, indent=1)
print(json.dumps(res))
