#!/usr/bin/env python3
"""Per-kernel times of the step on per-env clocks (G clock groups in one handle, every env served) against lock-step:
python tools/experiments/staggered_kernels.py [G]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 65536
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
          fc_max_power=100.0, fcev_permeate=0.01)
v = chub.VecChargingHub(n, seed=1, **kw)
lib, h = v._lib, v._h
st = multi_gpu.Stream(0)
acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
for b, a in enumerate(acts):
    v.random_actions_device(a.ptr, 123, b, st.ptr)
packed = multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4)
obs = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)
rew = multi_gpu.DeviceBuffer(n * 4)
done = multi_gpu.DeviceBuffer(n)
grp = np.arange(n) // (n // G)
masks = [np.ascontiguousarray(grp == g, dtype=np.uint8) for g in range(G)]


def day(staggered, label):
    t_grp = v.env_clocks()[::n // G].copy()
    for rep in range(2):
        if rep == 1:
            v.profile_begin(96, every=5)
        for i in range(96 * 5):
            v.step_device_packed(acts[i % 4].ptr, packed.ptr, stream=st.ptr)
            t_grp = (t_grp + 1) % 96
            for g in np.nonzero(t_grp == 0)[0]:
                if staggered:
                    check(lib.chub_reset_envs_device(h, masks[g].ctypes.data, None, None, obs.ptr, st.ptr))
                else:
                    v.reset_device(obs.ptr, stream=st.ptr)
                    break
    a, b, k = v.profile_end()
    print("%s: slot %.2f us, env %.2f us over %d steps" % (label, a / k * 1e3, b / k * 1e3, k))


v.reset_device(obs.ptr, stream=st.ptr)
day(False, "lock-step")
v.reset_device(obs.ptr, stream=st.ptr)
for k in range(1, 96 * (G - 1) // G + 1):
    m = np.ascontiguousarray(grp * 96 // G >= k, dtype=np.uint8)
    check(lib.chub_step_envs_device(h, m.ctypes.data, acts[k % 4].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))
st.sync()
day(True, "%d clocks" % G)
v.close()
