#!/usr/bin/env python3
"""Where the life of a tail workgroup goes in the reference-exact COMPAT mode (k_env<.., COMPAT> behind k_slot_walk2): the tail half of
tools/experiments/phase_stamps.py on a COMPAT handle.  Measurement build, on a GPU box:
    CHUB_LIB=$PWD/charginghub-env_amd/libchub_t.so python tools/experiments/phase_stamps_compat.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import numpy as np

import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu

n = int(os.environ.get("AB_ENVS", "65536"))
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0,
          fcev_permeate=0.01)
v = chub.VecChargingHub(n, seed=1, rng="compat", **kw)
v.compat_replay_constructor()
lib = v._lib
A, D = v.act_dim, v.obs_dim
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
nb_env = (n + 255) // 256
st_slot, st_env = multi_gpu.DeviceBuffer(16 * 8), multi_gpu.DeviceBuffer(nb_env * 16 * 8)
st_env.from_host(np.zeros(nb_env * 16, dtype=np.uint64))
v.reset_device(obs.ptr, days.ptr, zs[0].ptr, stream=st.ptr)
for t in range(40):
    v.step_device(acts[t & 1].ptr, obs.ptr, rew.ptr, done.ptr, d_exo_z=zs[t & 1].ptr, stream=st.ptr)
st.sync()
lib.chub_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.chub_debug_stamps.restype = None
lib.chub_debug_stamps(None, st_env.ptr)
acc = []
for t in range(40, 64):
    v.step_device(acts[t & 1].ptr, obs.ptr, rew.ptr, done.ptr, d_exo_z=zs[t & 1].ptr, stream=st.ptr)
    st.sync()
    acc.append(st_env.to_host(np.uint64, (nb_env, 16)).astype(np.int64))
lib.chub_debug_stamps(None, None)
e = np.stack(acc)
ok = (e[:, :, :10] > 0).all(axis=(0, 2))
e = e[:, ok]
tick = ((e[:, :, 7] - e[:, :, 0]).sum() / ((e[:, :, 9] - e[:, :, 8]).sum() / 100.0))
names = ["entry -> kernel arguments here", "-> load burst issued", "-> loads landed, rows parked (barrier)", "-> first half (exogenous, forecourt)",
         "-> second half (H2, money, observation)", "-> rows flushed", "-> stores drained"]
de = np.diff(e[:, :, :8], axis=2).astype(np.float64)
life = (e[:, :, 7] - e[:, :, 0]).astype(np.float64)
print("COMPAT, %d envs x %s: %d tail workgroups with a complete set of stamps of %d; shader clock %.0f ticks per us; life mean %.2f us" % (
    n, kw["station_list"], ok.sum(), ok.size, tick, life.mean() / tick))
for i, nm in enumerate(names):
    print("  %-50s %6.2f us  (%4.1f %%)" % (nm, de[:, :, i].mean() / tick, 100 * de[:, :, i].mean() / life.mean()))
start = (e[:, :, 8] - e[:, :, 8].min(axis=1, keepdims=True)).astype(np.float64) / 100.0
print("  workgroup entry times (shared clock): median %.2f us, last %.2f us after the first; last exit %.2f us" % (
    np.median(start), start.max(axis=1).mean(), ((e[:, :, 9].max(axis=1) - e[:, :, 8].min(axis=1)) / 100.0).mean()))
v.close()
