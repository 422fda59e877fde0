#!/usr/bin/env python3
"""Per-kernel step times of one libchub build, for A/B comparisons INSIDE one GPU-box call (box-to-box variation is
+-3 %, far more than most kernel changes):  for v in a b a b; do python3 tools/ab_step_times.py --lib $PWD/charginghub-env_amd/libchub_$v.so; done"""
import argparse, os, sys
sys.path.insert(0, ".")
# every knob is a flag (the AB_* environment variables of rounds 4-5 still work as defaults): under `rocprofv3 ... --` the command must be the
# program itself -- `python3 tools/ab_step_times.py --order dispatch` -- never `env VAR=... python3 ...` (an exec after the profiler's preload
# has initialised the GPU, which this pool forbids)
ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["c4", "c5", "c2"], default=os.environ.get("AB_CONFIG", "c4"))
ap.add_argument("--envs", type=int, default=int(os.environ["AB_ENVS"]) if "AB_ENVS" in os.environ else None)
ap.add_argument("--steps", type=int, default=int(os.environ["AB_STEPS"]) if "AB_STEPS" in os.environ else None)
ap.add_argument("--warm", type=int, default=int(os.environ["AB_WARM"]) if "AB_WARM" in os.environ else None)
ap.add_argument("--tile", default=os.environ.get("AB_TILE", "auto"))
ap.add_argument("--fused", default=os.environ.get("AB_FUSED", "auto"))
ap.add_argument("--order", default=os.environ.get("AB_ORDER", "auto"))
ap.add_argument("--lib", default=None, help="another build of libchub.so (the same as CHUB_LIB=...)")
ap.add_argument("--no-events", action="store_true", help="no per-kernel dispatch timestamps (chub_profile_*: hipExtLaunchKernelGGL with start / stop events)")
ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replays at the end (counter passes: the eager steps are what is counted)")
args = ap.parse_args()
if args.lib:
    os.environ["CHUB_LIB"] = args.lib  # (read by charginghub_env_amd._lib.lib_path at load_library(): before any HIP call)
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
kw = dict(station_list=[20,25], station_type_list=["fast","slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
n, STEPS, WARM = 65536, 1920, 960
if args.config == "c5":  # 262 144 envs x [32, 32]: the working set beyond the Infinity Cache
    n, STEPS, WARM = 262144, 480, 192
    kw.update(station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3)
if args.config == "c2":  # 4096 envs x [16, 0]: the single-launch step
    n = 4096
    kw.update(station_list=[16, 0], fcev_permeate=0.0)
n = args.envs if args.envs is not None else n
STEPS = args.steps if args.steps is not None else STEPS
WARM = args.warm if args.warm is not None else WARM
v = chub.VecChargingHub(n, seed=1, tile=args.tile, fused_step=args.fused, work_order=args.order, **kw)
A = v.act_dim
acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
for b, a in enumerate(acts): v.random_actions_device(a.ptr, 123, b, 0)
packed = multi_gpu.DeviceBuffer(n * 15 * 4); obs0 = multi_gpu.DeviceBuffer(n * 13 * 4); D2 = v.obs_dim + 2
for i in range(WARM):
    if i % 96 == 0: v.reset_device(obs0.ptr)
    v.step_device_packed(acts[i%4].ptr, packed.ptr)
v.sync()
if not args.no_events: v.profile_begin(STEPS, every=2)
import time
t0 = time.perf_counter()
for i in range(STEPS):
    if i % 96 == 0: v.reset_device(obs0.ptr)
    v.step_device_packed(acts[i%4].ptr, packed.ptr)
a, b, k = v.profile_end() if not args.no_events else (0.0, 0.0, 1)
v.sync()
dt = time.perf_counter() - t0
import numpy as np
# the same steps as hipGraph replays of two episodes (what bench.py times at N = 1): microseconds per step without the host
graph_us = float("nan")
if not args.no_graph:
    st = multi_gpu.Stream(0)
    pk2 = [multi_gpu.DeviceBuffer(n * D2 * 4) for _ in range(2)]
    def two_days():
        for i in range(192):
            if i % 96 == 0: v.reset_device(obs0.ptr, stream=st.ptr)
            v.step_device_packed(acts[i%4].ptr, pk2[i&1].ptr, stream=st.ptr)
    v.sync(); two_days(); st.sync()
    v.graph_begin(st.ptr); two_days(); g = v.graph_end(st.ptr)
    v.graph_launch(g, st.ptr); st.sync()
    reps = 20 if n <= 65536 else 5
    t0 = time.perf_counter()
    for _ in range(reps): v.graph_launch(g, st.ptr)
    st.sync()
    graph_us = (time.perf_counter() - t0) / (reps * 192) * 1e6
    v.graph_destroy(g)
chk = float(packed.to_host(np.float32, (n, D2)).astype(np.float64).sum())  # same seeds, same result whatever the build
print(os.environ.get("CHUB_LIB","")[-12:], args.tile, args.fused, args.order, n, kw["station_list"], "slot_us %.2f env_us %.2f step_us %.2f graph_us %.2f  checksum %.6f" % (a/k*1e3, b/k*1e3, dt/STEPS*1e6, graph_us, chk))
v.close()
