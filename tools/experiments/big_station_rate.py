#!/usr/bin/env python3
"""Step times on hubs whose stations are walked in chunks (k_slot_unit_any; a parity-grade path): device-resident actions, call by call.
usage (GPU box, repo root): python3 tools/experiments/big_station_rate.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
base = dict(station_type_list=["fast", "slow"], hydro_prod_rate=2000.0, hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=100.0, fcev_permeate=0.02)
for rng in ("philox", "compat"):
    for n, piles in ((1024, [300, 270]), (4096, [300, 20]), (256, [1000, 1000]), (64, [4096, 3])):
        v = chub.VecChargingHub(n, seed=5, rng=rng, station_list=piles, **base)
        acts = multi_gpu.DeviceBuffer(n * v.act_dim * 4)
        v.random_actions_device(acts.ptr, 9, 0, 0)
        packed = multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4)
        obs0 = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)
        z = multi_gpu.DeviceBuffer(n * 3 * 8)
        days = multi_gpu.DeviceBuffer(n * 2 * 4)
        z.from_host(np.zeros((n, 3)))
        days.from_host(np.zeros((n, 2), dtype=np.int32))
        kz = dict(d_exo_z=z.ptr) if rng == "compat" else {}
        def day():
            if rng == "compat":
                v.reset_device(obs0.ptr, d_exo_days=days.ptr, d_exo_z=z.ptr)
            else:
                v.reset_device(obs0.ptr)
            for i in range(96):
                v.step_device_packed(acts.ptr, packed.ptr, **kz)
        day(); v.sync()
        t0 = time.perf_counter(); day(); day(); v.sync()
        us = (time.perf_counter() - t0) / 192 * 1e6
        print("%-7s %5d envs x %-12s %8.1f us per step  %7.2f M env-steps/s  %8.1f M slot-steps/s" % (rng, n, piles, us, n / us, n * sum(piles) / us), flush=True)
        v.close()
