#!/bin/bash
# usage (on the GPU box, from the repo root): tools/all_configs.sh <tag> -- bench.py on every BASELINE.json config that fits one GPU,
# one JSON line each, into gpurun_out/round6_<tag>_all_configs_1gpu.json (copy into profiles/ afterwards)
TAG=${1:-vX}
OUT=gpurun_out/round6_${TAG}_all_configs_1gpu.json
: > $OUT
for c in c2 c3 c4 c5; do
  python3 bench.py --config $c --steps 1920 --warmup 384 --no-cpu-baseline --no-c5 2>/dev/null | tail -1 >> $OUT
done
# the per-GPU share of the weak-scaling job (configs[4] over 8 GPUs: 32 768 envs x [32,32] per GPU)
python3 bench.py --scaling weak --steps 1920 --warmup 384 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
# the multi-GPU call pattern on one GPU: communicator of one rank, gather to itself
python3 bench.py --force-comm --steps 1920 --warmup 384 --no-cpu-baseline --no-c5 2>/dev/null | tail -1 >> $OUT
python3 - $OUT <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    r = d["roofline"]
    print(d["config"]["workload"][:48], "| %.0f M/s | %.2f us/step | slot %.1f us env %.1f us | frac %.3f step %.3f | %s"
          % (d["value"] / 1e6, d["ms_per_step"] * 1e3, r["avg_launch_us"], r["env_kernel_avg_launch_us"], r["frac"],
             d["roofline_step"]["frac"], d["config"]["collective"][:30]))
PY
# the same workloads with one bit per pile as the action input (chub_step_bits_device), next to the float rows
for c in c2 c3 c4 c5; do python3 tools/bits_device_rate.py $c; done > gpurun_out/round6_${TAG}_packed_actions.txt 2>&1
cat gpurun_out/round6_${TAG}_packed_actions.txt
