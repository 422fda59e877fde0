#!/usr/bin/env python3
"""The reference-exact COMPAT mode as a batch: microseconds per step (device-resident inputs, call by call), the station kernels' and the
tail kernel's share (dispatch timestamps), and a checksum of the end state for A/B runs of two builds (CHUB_LIB=...):
    AB_ENVS=65536 python tools/compat_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu

n = int(os.environ.get("AB_ENVS", "65536"))
piles = [int(x) for x in os.environ.get("AB_PILES", "20,25").split(",")]
kw = dict(station_list=piles, station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
# AB_SLOT=wave: one kernel per station; AB_WALK=off: every step walks for itself in front of its slot pass (k_compat_walk + k_slot_split2 + k_env)
v = chub.VecChargingHub(n, seed=1, rng="compat", slot_kernel=os.environ.get("AB_SLOT", "auto"), walk_ahead=os.environ.get("AB_WALK", "auto"), **kw)
v.compat_replay_constructor()
D, A = v.obs_dim, v.act_dim
st = multi_gpu.Stream(0)
rs = np.random.RandomState(1)
acts, zs = [], []
for b in range(2):
    a = multi_gpu.DeviceBuffer(n * A * 4)
    v.random_actions_device(a.ptr, 123, b, st.ptr)
    z = multi_gpu.DeviceBuffer(n * 3 * 8)
    z.from_host(rs.normal(size=(n, 3)), st.ptr)
    acts.append(a)
    zs.append(z)
days = multi_gpu.DeviceBuffer(n * 2 * 4)
days.from_host(np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32), st.ptr)
obs, rew, done = multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)


def day():
    v.reset_device(obs.ptr, days.ptr, zs[0].ptr, stream=st.ptr)
    for t in range(96):
        v.step_device(acts[t & 1].ptr, obs.ptr, rew.ptr, done.ptr, d_exo_z=zs[t & 1].ptr, stream=st.ptr)
    st.sync()


day()
v.profile_begin(96, every=1)
t0 = time.perf_counter()
day()
dt = time.perf_counter() - t0
a, b, k = v.profile_end()
o = obs.to_host(np.float32, (n, D), st.ptr).astype(np.float64)
sl = np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1).astype(np.float64)
print(os.environ.get("CHUB_LIB", "")[-12:], os.environ.get("AB_SLOT", "auto"), n, piles, "step_us %.1f  slot pass (+ walks) %.1f  tails (+ walks) %.1f   (%.0f M env-steps/s)   checksum %.9f %.6f" % (
    dt / 96 * 1e6, a / k * 1e3, b / k * 1e3, n * 96 / dt / 1e6, o.sum(), np.nansum(sl)))
v.close()
