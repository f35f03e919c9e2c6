#!/bin/bash
# GPU box: the bench line (cfg 2) and the secondary workloads
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err || exit 2
python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_final.json').read().strip().splitlines()[-1])
print('cfg2', d['ms_per_step'], d['value'], 'xattn', d['cross_attention']['fwd_bwd_tflops'], 'packed', d['ragged_masks'].get('packed_ms_per_step'), d['ragged_masks']['ms_per_step'])
print({k: v['frac'] for k, v in d['roofline']['achieved_by_class'].items()})"
for w in cfg4 cfg5 cfg5_fp8; do
  timeout -k 10 200 python bench.py --workload $w --steps 30 --no-cpu-baseline > gpurun_out/r4_bench_$w.json 2> gpurun_out/r4_bench_$w.err || exit 3
  python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d['roofline']['device_ms_per_step_by_class'])"
done
