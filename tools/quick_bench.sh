#!/bin/bash
# usage: tools/quick_bench.sh "ENV=.. ENV2=.." [steps]   -> one line with slot/env kernel times
env $1 timeout -k 10 200 python bench.py --steps ${2:-192} --warmup 96 --no-cpu-baseline ${3:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'slot_us', round(d['roofline']['avg_launch_us'],1), 'env_us', round(d['roofline']['env_kernel_avg_launch_us'],1), 'ms/step', round(d['ms_per_step'],4), 'Msteps/s', round(d['value']/1e6,1))"
