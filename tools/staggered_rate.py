#!/usr/bin/env python3
"""Step rate of ONE handle whose envs run on G different clocks (per-env clocks, include/chub.h), device-resident, against the
same handle in lock-step:  python tools/staggered_rate.py [G]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(os.environ.get("AB_ENVS", "65536"))
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
          fc_max_power=100.0, fcev_permeate=0.01)
v = chub.VecChargingHub(n, seed=1, **kw)
lib, h = v._lib, v._h
acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
for b, a in enumerate(acts):
    v.random_actions_device(a.ptr, 123, b, 0)
packed = multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4)
obs = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)
rew = multi_gpu.DeviceBuffer(n * 4)
done = multi_gpu.DeviceBuffer(n)
grp = np.arange(n) // (n // G)


def run(steps, staggered):
    t_grp = v.env_clocks()[::n // G].copy()  # the host keeps one clock per group (per-env bookkeeping would cost more than a step)
    masks = [np.ascontiguousarray(grp == g, dtype=np.uint8) for g in range(G)]
    t0 = time.perf_counter()
    for i in range(steps):
        v.step_device_packed(acts[i % 4].ptr, packed.ptr)
        t_grp = (t_grp + 1) % 96
        for g in np.nonzero(t_grp == 0)[0]:
            if staggered:
                check(lib.chub_reset_envs_device(h, masks[g].ctypes.data, None, None, obs.ptr, None))
            else:
                v.reset_device(obs.ptr)
                break
    v.sync()
    return (time.perf_counter() - t0) / steps * 1e6


v.reset_device(obs.ptr)
print("lock-step: %.1f us per step (%d clocks)" % (run(960, False), v.clock_groups))
v.reset_device(obs.ptr)
for k in range(1, 96 * (G - 1) // G + 1):  # head starts: group g ends up g * 96 / G slots ahead
    m = np.ascontiguousarray(grp * 96 // G >= k, dtype=np.uint8)
    check(lib.chub_step_envs_device(h, m.ctypes.data, acts[k % 4].ptr, None, obs.ptr, rew.ptr, done.ptr, None))
v.sync()
print("clocks:", sorted(set(v.env_clocks().tolist())), v.clock_groups)
us = run(960, True)
print("%d clocks in one handle: %.1f us per step = %.0f M env-steps/s (%d clocks at the end)" % (G, us, n / us, v.clock_groups))

# the same schedule as hipGraph replays: one period (96 steps + G masked resets) captured on per-env clocks, replayed 10 times
st = multi_gpu.Stream(0)
periods = 1 if (96 + G) % 2 == 0 else 2
masks = [np.ascontiguousarray(grp == g, dtype=np.uint8) for g in range(G)]
t_grp0 = v.env_clocks()[::n // G].copy()
v.sync()
v.graph_begin(st.ptr)
t_grp = t_grp0.copy()
for i in range(96 * periods):
    v.step_device_packed(acts[i % 4].ptr, packed.ptr, stream=st.ptr)
    t_grp = (t_grp + 1) % 96
    for g in np.nonzero(t_grp == 0)[0]:
        check(lib.chub_reset_envs_device(h, masks[g].ctypes.data, None, None, obs.ptr, st.ptr))
graph = v.graph_end(st.ptr)
v.graph_launch(graph, st.ptr)
st.sync()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    v.graph_launch(graph, st.ptr)
st.sync()
us = (time.perf_counter() - t0) / (reps * 96 * periods) * 1e6
print("%d clocks in one handle, hipGraph replays of one period: %.1f us per step = %.0f M env-steps/s (%d clocks at the end)"
      % (G, us, n / us, v.clock_groups))
v.graph_destroy(graph)
v.close()
