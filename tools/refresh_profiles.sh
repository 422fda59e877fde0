#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/refresh_profiles.sh v3
# Writes gpurun_out/profiles_<tag>/: the FETCH_SIZE / WRITE_SIZE / SQ counter passes (each --pmc set in its own run) summarised
# into one JSON, then the default bench line and the rocprofv3 kernel-trace stats of the same command.
# Copy the files into profiles/ (tracked) afterwards.
set -e
TAG=${1:-vX}
ROUND=${ROUND:-round6}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 192 --warmup 96 --no-cpu-baseline --no-events --no-sustained"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
echo "write pass done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
echo "sq pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_tcc.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_tcp --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_tcp.err
echo "cache passes done"
# the same two traffic passes on the working set that does not fit the Infinity Cache (bench.py's roofline_c5 block)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_c5 --output-format csv -- python3 bench.py --config c5 $ARGS > /dev/null 2> $OUT/pmc_fetch_c5.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_c5 --output-format csv -- python3 bench.py --config c5 $ARGS > /dev/null 2> $OUT/pmc_write_c5.err
echo "c5 traffic passes done"
python3 - "$OUT" "$TAG" "$ROUND" "$ARGS" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, tag, rnd, bench_args = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
def collect(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc
def pick(acc, key):
    for name in acc:
        if any(k in name for k in key):
            return {c: sum(v) / len(v) for c, v in acc[name].items()}
    return {}
fe, wr, sq = collect(out + "/pmc_fetch"), collect(out + "/pmc_write"), collect(out + "/pmc_sq")
for extra in ("/pmc_tcc", "/pmc_tcp"):
    for name, d in collect(out + extra).items():
        for c, v in d.items():
            sq[name][c] = v
import ctypes, os
sys.path.insert(0, os.getcwd())
import charginghub_env_amd as chub
build_id = chub.load_library().chub_build_id().decode()
# gfx950 corrections (MI355X_MICROARCH.md, HBM section; confirmed in round 1 with tools/microbench/copy4.hip): FETCH_SIZE
# reports half of the bytes fetched (128-byte requests tallied at 64 bytes), WRITE_SIZE is exact; both in KiB
FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / TCC_* / TCP_* (separate passes), mean per dispatch over the step kernels "
               "of bench.py " + bench_args + " at 65536 envs x [20,25] (the timed steps of such a run go out as hipGraph replays: the sampled "
               "dispatches are graph-launched kernels); KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950",
       "build_id": build_id, "envs": 65536, "hub": [20, 25],
       "calibration": {"fetch_factor": FETCH_FACTOR, "write_factor": WRITE_FACTOR}}
KEYS = (("k_slot", ("k_slot_packed<256, 2, false, false",)), ("k_env", ("k_env<false",)))
for label, key in KEYS:
    f, w = pick(fe, key).get("FETCH_SIZE", 0.0), pick(wr, key).get("WRITE_SIZE", 0.0)
    res[label] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                  "traffic_bytes_per_launch": (f * FETCH_FACTOR + w * WRITE_FACTOR) * 1024.0}
    res[label + "_sq_counters_per_dispatch"] = pick(sq, key)
json.dump(res, open("%s/%s_%s_pmc_traffic.json" % (out, rnd, tag), "w"), indent=1)
print(json.dumps(res["k_slot"]))
fe5, wr5 = collect(out + "/pmc_fetch_c5"), collect(out + "/pmc_write_c5")
res5 = {"note": "the same FETCH_SIZE / WRITE_SIZE passes on bench.py --config c5 (262144 envs x [32,32]: 134 MB of slot state, beyond the "
                "256 MB Infinity Cache together with the action batches): what bench.py's roofline_c5.traffic reports",
        "build_id": build_id, "envs": 262144, "hub": [32, 32], "calibration": res["calibration"]}
KEYS5 = (("k_slot", ("k_slot_packed<512, 4, false, false",)), ("k_env", ("k_env<false",)))  # this working set runs on the second tile
for label, key in KEYS5:
    f, w = pick(fe5, key).get("FETCH_SIZE", 0.0), pick(wr5, key).get("WRITE_SIZE", 0.0)
    res5[label] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": (f * FETCH_FACTOR + w * WRITE_FACTOR) * 1024.0}
json.dump(res5, open("%s/%s_%s_pmc_traffic_c5.json" % (out, rnd, tag), "w"), indent=1)
print(json.dumps(res5["k_slot"]))
PY
# (2) the reference-exact COMPAT mode at the headline size (tools/compat_rate.py: 65 536 envs x [20, 25], 2 days call by call): kernel-trace
# stats + one counter set per pass for k_slot_walk2 (the slot pass beside the next step's stream walks) / k_env (the tails), to check roofline_compat against
rocprofv3 --kernel-trace --stats -d $OUT/ktc --output-format csv -- python3 tools/compat_rate.py > $OUT/compat_rate_under_rocprof.txt 2> $OUT/ktc.err
cp $(find $OUT/ktc -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_${TAG}_kernel_stats_compat.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_compat --output-format csv -- python3 tools/compat_rate.py > /dev/null 2> $OUT/pmc_fetch_compat.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_compat --output-format csv -- python3 tools/compat_rate.py > /dev/null 2> $OUT/pmc_write_compat.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq_compat --output-format csv -- python3 tools/compat_rate.py > /dev/null 2> $OUT/pmc_sq_compat.err
echo "compat passes done"
python3 - "$OUT" "$TAG" "$ROUND" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, tag, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
def collect(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc
def pick(acc, key):
    for name in acc:
        if key in name:
            return {c: sum(v) / len(v) for c, v in acc[name].items()}
    return {}
sys.path.insert(0, os.getcwd())
import charginghub_env_amd as chub
fe, wr, sq = collect(out + "/pmc_fetch_compat"), collect(out + "/pmc_write_compat"), collect(out + "/pmc_sq_compat")
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes), mean per dispatch over the COMPAT step kernels of "
               "tools/compat_rate.py (65536 envs x [20,25], reference streams, device-resident actions and normals, call by call); KiB; FETCH_SIZE "
               "doubled as MI355X_MICROARCH.md prescribes for gfx950",
       "build_id": chub.load_library().chub_build_id().decode(), "envs": 65536, "hub": [20, 25], "mode": "compat",
       "calibration": {"fetch_factor": 2.0, "write_factor": 1.0}}
# (the split step walks two steps ahead: k_slot_walk2 + k_env<.., COMPAT> per step; k_compat_walk<false> only in front of a day's first step)
for label, key in (("k_slot_walk2", "k_slot_walk2"), ("k_env_compat", "k_env<false, 0"), ("k_compat_walk", "k_compat_walk<false>")):
    f, w = pick(fe, key).get("FETCH_SIZE", 0.0), pick(wr, key).get("WRITE_SIZE", 0.0)
    res[label] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": (f * 2.0 + w) * 1024.0}
    res[label + "_sq_counters_per_dispatch"] = pick(sq, key)
# the whole COMPAT step as bench.py's roofline_compat prices it: both launches of a lock-step step
res["traffic_bytes_per_step"] = res["k_slot_walk2"]["traffic_bytes_per_launch"] + res["k_env_compat"]["traffic_bytes_per_launch"]
json.dump(res, open("%s/%s_%s_pmc_compat.json" % (out, rnd, tag), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.startswith("k_") and not k.endswith("dispatch")}))
PY
# the traffic summaries go where bench.py looks for them (profiles/, matched by build id), so that the bench lines written next
# carry `roofline.traffic` of this very build
cp $OUT/${ROUND}_${TAG}_pmc_traffic.json $OUT/${ROUND}_${TAG}_pmc_traffic_c5.json $OUT/${ROUND}_${TAG}_pmc_compat.json profiles/
python3 bench.py > $OUT/${ROUND}_${TAG}_bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 400 $OUT/${ROUND}_${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --no-sustained > $OUT/${ROUND}_${TAG}_bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_${TAG}_kernel_stats.csv
echo "kernel trace done"
# ---- round 5: the two workloads whose fractions the bench line quotes without a rocprof summary so far
# (1) C5 as the bench's own workload (262 144 envs x [32, 32]): kernel-trace stats, to check roofline_c5.avg_launch_us against
rocprofv3 --kernel-trace --stats -d $OUT/kt5 --output-format csv -- python3 bench.py --config c5 --no-cpu-baseline --no-sustained > $OUT/${ROUND}_${TAG}_bench_c5_under_rocprof.json 2> $OUT/kt5.err
cp $(find $OUT/kt5 -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_${TAG}_kernel_stats_c5.csv
echo "c5 kernel trace done"
# (round 6) C2, the small batch: every step a launch (k_step_tailwave) and spans of steps in one launch (k_steps_piped: one launch = up to 96 steps)
rocprofv3 --kernel-trace --stats -d $OUT/kt2 --output-format csv -- python3 tools/span_rate.py --config c2 --days 6 > $OUT/${ROUND}_${TAG}_span_rate_c2_under_rocprof.txt 2> $OUT/kt2.err
cp $(find $OUT/kt2 -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_${TAG}_kernel_stats_c2.csv
rm -rf $OUT/kt2
echo "c2 kernel trace done"
rm -rf $OUT/kt $OUT/kt5 $OUT/ktc $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_tcc $OUT/pmc_tcp $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 $OUT/pmc_fetch_compat $OUT/pmc_write_compat $OUT/pmc_sq_compat
ls $OUT
