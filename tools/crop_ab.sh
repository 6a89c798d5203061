#!/bin/bash
# class-2 crops of the certified argmax (TTUP_CERT_SMALL=0: off; read once per process): the bench regimes on ONE box, alternating
for sm in 0 1 0 1; do
  TTUP_CERT_SMALL=$sm python bench.py --no-cpu-baseline > gpurun_out/bench_small$sm.json 2> gpurun_out/bench_small$sm.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_small$sm.json'))
n=d['noise_weights_fps']; v=d['varied_content_fps']
print('small $sm', d['value'], 'varied', v['value'], v['crops_per_heatmap'], 'exact', d['exact_windows_fps']['value'], 'noise', n['value'], n['crops_per_heatmap'], n.get('small_core_crop_share'), v.get('small_core_crop_share'), d['exact_windows_fps'].get('small_core_crop_share'), 'cnn', d['cnn_only_fps']['value'], 'hub', d['hub_clip_fps']['value'], d['hub_clip_fps_256']['value'])
"
done
