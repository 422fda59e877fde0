#!/usr/bin/env python3
"""Where a workgroup's life goes, in both step kernels: mean time between the s_memtime stamps of a measurement build.

    make -C charginghub-env_amd/csrc KFLAGS=-DCHUB_TRACE=1 && cp charginghub-env_amd/libchub.so charginghub-env_amd/libchub_t.so
    make -C charginghub-env_amd/csrc
    CHUB_LIB=$PWD/charginghub-env_amd/libchub_t.so python tools/experiments/phase_stamps.py      # on a GPU box; AB_CONFIG=c5 / c2 as ab_step_times.py

A stamp waits for scalar results only: a phase's share is where the workgroup's first wave stood (issue + the waits it could not overlap).
The counter is the shader clock; its rate is calibrated here against the kernels' own dispatch timestamps."""
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
if os.environ.get("AB_CONFIG") == "c5":
    n = int(os.environ.get("AB_ENVS", "262144"))
    kw.update(station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3)
v = chub.VecChargingHub(n, seed=1, fused_step="off", tile=os.environ.get("AB_TILE", "auto"), **kw)
lib = v._lib
A, D = v.act_dim, v.obs_dim
acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
for b, a in enumerate(acts):
    v.random_actions_device(a.ptr, 123, b, 0)
packed, obs0 = multi_gpu.DeviceBuffer(n * (D + 2) * 4), multi_gpu.DeviceBuffer(n * D * 4)
S = v.n_slots
tile = 2048 if (os.environ.get("AB_TILE") == "large" or (os.environ.get("AB_TILE", "auto") == "auto" and n * S >= 10 << 20)) else 512
nb_slot = (n + tile // S - 1) // (tile // S)
nb_env = (n + 255) // 256
nb_lvl = (3 * n + 255) // 256  # the level workgroups of k_env (next step's draws) follow the tail workgroups in its grid
st_slot, st_env = multi_gpu.DeviceBuffer(nb_slot * 16 * 8), multi_gpu.DeviceBuffer((nb_env + nb_lvl) * 16 * 8)
st_slot.from_host(np.zeros(nb_slot * 16, dtype=np.uint64))
st_env.from_host(np.zeros((nb_env + nb_lvl) * 16, dtype=np.uint64))
v.reset_device(obs0.ptr)
for i in range(40):
    v.step_device_packed(acts[i % 4].ptr, packed.ptr)
v.sync()
lib.chub_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.chub_debug_stamps.restype = None
lib.chub_debug_stamps(st_slot.ptr, st_env.ptr)
acc_s, acc_e = [], []
for i in range(40, 40 + 24):
    v.step_device_packed(acts[i % 4].ptr, packed.ptr)
    v.sync()
    acc_s.append(st_slot.to_host(np.uint64, (nb_slot, 16)).astype(np.int64))
    acc_e.append(st_env.to_host(np.uint64, (nb_env + nb_lvl, 16)).astype(np.int64))
lib.chub_debug_stamps(None, None)
v.profile_begin(64, every=1)
for i in range(64):
    v.step_device_packed(acts[i % 4].ptr, packed.ptr)
x, y, k = v.profile_end()
slot_us, env_us = x / k * 1e3, y / k * 1e3
s, e_all = np.stack(acc_s), np.stack(acc_e)  # [steps, blocks, words]
# (shader-clock stamps + two stamps of the 100 MHz clock all XCDs share)
e, lv = e_all[:, :nb_env], e_all[:, nb_env:]
ok_s, ok_e = (s[:, :, :14] > 0).all(axis=(0, 2)), (e[:, :, :10] > 0).all(axis=(0, 2))
print("%d envs x %s; dispatch timestamps: slot kernel %.2f us, tail kernel %.2f us" % (n, kw["station_list"], slot_us, env_us))
print("workgroups with a complete set of stamps: %d of %d (slot kernel), %d of %d (tail kernel)" % (ok_s.sum(), ok_s.size, ok_e.sum(), ok_e.size))
s, e = s[:, ok_s], e[:, ok_e]
# shader-clock ticks per us: a workgroup's entry -> last common stamp in both clocks (the shared clock has 10 ns steps)
tick = ((s[:, :, 10] - s[:, :, 0]).sum() / ((s[:, :, 13] - s[:, :, 12]).sum() / 100.0))
tick_e = ((e[:, :, 7] - e[:, :, 0]).sum() / ((e[:, :, 9] - e[:, :, 8]).sum() / 100.0))
print("shader clock: %.0f ticks per us in the slot kernel, %.0f in the tail kernel" % (tick, tick_e))
names_s = ["entry -> first loads issued", "-> state words here, row reads issued", "-> ballots done (at barrier 1)", "barrier 1", "-> rows consumed, sums added (at barrier 2)",
           "barrier 2", "-> admission done, state stored (at barrier 3)", "barrier 3", "-> new cars done", "barrier 4", "-> records stored (last wave)"]
d = np.diff(s[:, :, :12], axis=2).astype(np.float64)
life = (s[:, :, 11] - s[:, :, 0]).astype(np.float64)
print("slot kernel: %d workgroups, life of a workgroup mean %.2f us (min %.2f, max %.2f)" % (nb_slot, life.mean() / tick, life.min() / tick, life.max() / tick))
for i, nm in enumerate(names_s):
    print("  %-50s %6.2f us  (%4.1f %%)" % (nm, d[:, :, i].mean() / tick, 100 * d[:, :, i].mean() / life.mean()))
start = (s[:, :, 12] - s[:, :, 12].min(axis=1, keepdims=True)).astype(np.float64) / 100.0
print("  workgroup entry times (shared clock): median %.2f us, 90 %% %.2f us, last %.2f us after the first; last exit %.2f us" % (
    np.median(start), np.percentile(start, 90), start.max(axis=1).mean(), ((s[:, :, 13].max(axis=1) - s[:, :, 12].min(axis=1)) / 100.0).mean()))
names_e = ["entry -> kernel arguments here", "-> load burst issued", "-> loads landed, rows parked (barrier)", "-> first half (exogenous, forecourt)", "-> second half (H2, money, observation)",
           "-> rows flushed", "-> stores drained"]
de = np.diff(e[:, :, :8], axis=2).astype(np.float64)
life_e = (e[:, :, 7] - e[:, :, 0]).astype(np.float64)
print("tail kernel: %d workgroups, life mean %.2f us" % (nb_env, life_e.mean() / tick_e))
for i, nm in enumerate(names_e):
    print("  %-50s %6.2f us  (%4.1f %%)" % (nm, de[:, :, i].mean() / tick_e, 100 * de[:, :, i].mean() / life_e.mean()))
start_e = (e[:, :, 8] - e[:, :, 8].min(axis=1, keepdims=True)).astype(np.float64) / 100.0
print("  workgroup entry times (shared clock): median %.2f us, last %.2f us after the first; last exit %.2f us" % (
    np.median(start_e), start_e.max(axis=1).mean(), ((e[:, :, 9].max(axis=1) - e[:, :, 8].min(axis=1)) / 100.0).mean()))
t0 = e[:, :, 8].min(axis=1, keepdims=True)
print("  level workgroups (%d, next step's draws): entry median %.2f us / last %.2f us after the first tail workgroup's entry; exit median %.2f us, LAST EXIT %.2f us (tail workgroups' last exit: %.2f us)" % (
    nb_lvl, np.median((lv[:, :, 8] - t0) / 100.0), ((lv[:, :, 8] - t0) / 100.0).max(axis=1).mean(), np.median((lv[:, :, 9] - t0) / 100.0),
    ((lv[:, :, 9] - t0) / 100.0).max(axis=1).mean(), ((e[:, :, 9] - t0) / 100.0).max(axis=1).mean()))
v.close()
