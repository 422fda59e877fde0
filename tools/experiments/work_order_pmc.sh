#!/bin/bash
# usage (on the GPU box, from the repo root): tools/experiments/work_order_pmc.sh <outdir>
# What the XCD-aware work order (chub_options.work_order) changes in the memory system: counter passes of the same
# short run (65 536 envs x [20,25], bench.py --work-order ...) in both orders, one --pmc set per run, mean per dispatch of the two step kernels.
OUT=${1:-gpurun_out/work_order_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 96 --warmup 96 --no-cpu-baseline --no-events --no-c5 --no-bits --no-dropin"
# (bench.py --no-events as the vehicle, as in tools/refresh_profiles.sh.  Round 5 had tools/ab_step_times.py here and lost two counter passes to a
# run that never finished.  Round 6 looked for the cause (gpurun_out/r6e, r6f; tools/experiments/README.md): NOT an `env` hop in front of python --
# the records show the profiled command was `python3 tools/ab_step_times.py` itself, variables exported by the shell, and the same command with
# flags instead of variables timed out again on its FIRST pass.  Under --pmc the script completes with its dispatch timestamps alone
# (chub_profile_*: hipExtLaunchKernelGGL with start / stop events; --no-graph) and with its captured graph alone (--no-events); with BOTH in one
# process it did not come back in 3 of 5 tries (rounds 5 and 6).  Counter passes therefore take --no-events (what is counted are the eager / graph-launched kernels
# themselves); every pass under its own timeout, and the script stops at the first pass that fails)
for order in auto dispatch; do
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    sleep 3
    timeout -k 10 300 rocprofv3 --pmc $set -d $OUT/${order}_p$i --output-format csv -- python3 bench.py $ARGS --work-order $order > /dev/null 2> $OUT/${order}_p$i.err || { echo "pass $order $i failed: $set"; exit 1; }
    echo "pass $order $i done: $set"
  done <<'SETS'
TCC_HIT_sum TCC_MISS_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
FETCH_SIZE
WRITE_SIZE
SETS
done
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
def collect(pat):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(pat, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc
rows = {}
for order in ("auto", "dispatch"):
    acc = collect(out + "/" + order + "_p*/**/*counter_collection.csv")
    for name, d in acc.items():
        key = "k_slot_packed<256,2> (step)" if "k_slot_packed<256, 2, false, false" in name else ("k_env (step)" if "k_env<false" in name else None)
        if key is None:
            continue
        for c, v in d.items():
            rows.setdefault((key, c), {})[order] = (sum(v) / len(v), len(v))
print("%-30s %-36s %16s %16s %8s" % ("kernel", "counter (mean per dispatch)", "XCD-aware", "dispatcher's", "ratio"))
for (key, c), d in sorted(rows.items()):
    a, b = d.get("auto", (float("nan"), 0))[0], d.get("dispatch", (float("nan"), 0))[0]
    print("%-30s %-36s %16.1f %16.1f %8.3f" % (key, c, a, b, a / b if b else float("nan")))
PY
rm -rf $OUT/auto_p[0-9]*/ $OUT/dispatch_p[0-9]*/
cat $OUT/summary.txt
