#!/usr/bin/env python3
"""Device-resident step rate with the actions as float rows (chub_step_device) against one bit per pile + the two tail floats
(chub_step_bits_device: the packed slot kernel reads the bits themselves), same decisions, same results:
python tools/bits_device_rate.py [c4|c5|c3|c2]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
HUB = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
           fc_max_power=100.0, fcev_permeate=0.01)
n, kw = {"c2": (4096, dict(HUB, station_list=[16, 0], fcev_permeate=0.0)), "c3": (32768, HUB), "c4": (65536, HUB),
         "c5": (262144, dict(HUB, station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3))}[cfg]
NB = 8
res = {}
for form in ("floats", "bits"):
    v = chub.VecChargingHub(n, seed=1, **kw)
    lib, h = v._lib, v._h
    st = multi_gpu.Stream(0)
    A, D, W = v.act_dim, v.obs_dim, v.bit_words
    acts, bits, tails = [], [], []
    for b in range(NB):
        a = multi_gpu.DeviceBuffer(n * A * 4)
        v.random_actions_device(a.ptr, 123, b, st.ptr)
        if form == "floats":
            acts.append(a)
            continue
        hb, ht = v.pack_actions(a.to_host(np.float32, (n, A), st.ptr))
        a.free()
        db, dt = multi_gpu.DeviceBuffer(n * W * 8), multi_gpu.DeviceBuffer(n * 2 * 4)
        check(lib.chub_copy_to_device(0, db.ptr, hb.ctypes.data, n * W * 8, st.ptr))
        check(lib.chub_copy_to_device(0, dt.ptr, ht.ctypes.data, n * 2 * 4, st.ptr))
        st.sync()
        bits.append(db); tails.append(dt)
    obs, rew, done = multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)

    def step(i):
        if i % 96 == 0:
            v.reset_device(obs.ptr, stream=st.ptr)
        if form == "floats":
            check(lib.chub_step_device(h, acts[i % NB].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))
        else:
            check(lib.chub_step_bits_device(h, bits[i % NB].ptr, tails[i % NB].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))

    for i in range(192):
        step(i)
    st.sync()
    # graph replays of two episodes: the rate without the host in the way
    v.graph_begin(st.ptr)
    for i in range(192):
        step(i)
    g = v.graph_end(st.ptr)
    v.graph_launch(g, st.ptr)
    st.sync()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        v.graph_launch(g, st.ptr)
    st.sync()
    us = (time.perf_counter() - t0) / (reps * 192) * 1e6
    v.graph_destroy(g)
    # per-kernel day averages, calls back to back
    for rep in range(2):
        if rep == 1:
            v.profile_begin(96, every=5)
        for i in range(480):
            step(i)
    a_ms, b_ms, k = v.profile_end()
    chk = float(obs.to_host(np.float32, (n, D), st.ptr).astype(np.float64).sum()) + float(rew.to_host(np.float32, (n,), st.ptr).astype(np.float64).sum())
    res[form] = chk
    print("%s %-6s: %.2f us per step as graph replays = %.0f M env-steps/s; slot kernel %.2f us, tail kernel %.2f us; checksum %.6f"
          % (cfg, form, us, n / us, a_ms / k * 1e3, b_ms / k * 1e3, chk))
    v.close()
assert res["floats"] == res["bits"], "the two forms disagree"
