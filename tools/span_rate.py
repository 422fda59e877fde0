#!/usr/bin/env python3
"""Small batches (handles on the one-launch step): microseconds per step with every step a launch (as hipGraph replays: what bench.py times) against
spans of steps in ONE launch (chub_run_steps -> k_steps_fused; chub_options.span_steps), eager and as graph replays, + a checksum of the last block:
    python3 tools/span_rate.py --config c2"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, ".")
ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["c2", "c4k", "shard"], default="c2")
ap.add_argument("--days", type=int, default=20)
ap.add_argument("--envs", type=int, default=0, help="another batch size than the config's")
ap.add_argument("--tails", default="auto", help="chub_options.span_tails: auto / same_wave / own_wave")
args = ap.parse_args()
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
n = args.envs or {"c2": 4096, "c4k": 4096, "shard": 8192}[args.config]
fused = "auto"
if args.config == "c2":
    kw.update(station_list=[16, 0], fcev_permeate=0.0)
if args.config == "shard":
    fused = "on"  # (745 workgroups: beyond the default's 384; the 8-GPU shard of the headline job)
out = []
for span in ("off", "auto", 24):
    v = chub.VecChargingHub(n, seed=1, span_steps=span, span_tails=args.tails, fused_step=fused, **kw)
    D, A = v.obs_dim, v.act_dim
    st = multi_gpu.Stream(0)
    acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, 123, b, st.ptr)
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
    obs0 = multi_gpu.DeviceBuffer(n * D * 4)
    c_acts = (C.c_void_p * 4)(*[a.ptr for a in acts])
    c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
    run = lambda first, count: check(v._lib.chub_run_steps(v._h, None, c_acts, 4, c_packed, None, obs0.ptr, first, count, st.ptr))
    run(0, 192)
    st.sync()
    t0 = time.perf_counter()
    run(192, 96 * args.days)
    st.sync()
    eager_us = (time.perf_counter() - t0) / (96 * args.days) * 1e6
    i = 192 + 96 * args.days
    v.graph_begin(st.ptr)
    run(i, 192)
    g = v.graph_end(st.ptr)
    v.graph_launch(g, st.ptr)
    st.sync()
    reps = max(1, args.days // 2)
    t0 = time.perf_counter()
    for _ in range(reps):
        v.graph_launch(g, st.ptr)
    st.sync()
    graph_us = (time.perf_counter() - t0) / (reps * 192) * 1e6
    v.graph_destroy(g)
    chk = float(packed[1].to_host(np.float32, (n, D + 2), st.ptr).astype(np.float64).sum())
    print("%-6s n %d hub %s  span_steps=%-4s  eager %.2f us/step   graph replays %.2f us/step   fused %s   checksum %.6f" % (
        args.config, n, kw["station_list"], span, eager_us, graph_us, v.uses_fused_step, chk))
    v.close()
    st.destroy()
