#!/usr/bin/env python3
"""Per-kernel step times of one libchub build, for A/B comparisons INSIDE one GPU-box call (box-to-box variation is
+-3 %, far more than most kernel changes):  for v in a b a b; do CHUB_LIB=$PWD/charginghub-env_amd/libchub_$v.so python
tools/ab_step_times.py; done"""
import os, sys
sys.path.insert(0, ".")
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
n = int(os.environ.get("AB_ENVS", "65536"))
kw = dict(station_list=[20,25], station_type_list=["fast","slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
STEPS, WARM = int(os.environ.get("AB_STEPS", "1920")), int(os.environ.get("AB_WARM", "960"))
if os.environ.get("AB_CONFIG") == "c5":  # 262 144 envs x [32, 32]: the working set beyond the Infinity Cache
    n = int(os.environ.get("AB_ENVS", "262144"))
    kw.update(station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3)
    STEPS, WARM = 480, 192
if os.environ.get("AB_CONFIG") == "c2":  # 4096 envs x [16, 0]: the single-launch step
    n = int(os.environ.get("AB_ENVS", "4096"))
    kw.update(station_list=[16, 0], fcev_permeate=0.0)
v = chub.VecChargingHub(n, seed=1, tile=os.environ.get("AB_TILE", "auto"), fused_step=os.environ.get("AB_FUSED", "auto"),
                        work_order=os.environ.get("AB_ORDER", "auto"), **kw)
A = v.act_dim
acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
for b, a in enumerate(acts): v.random_actions_device(a.ptr, 123, b, 0)
packed = multi_gpu.DeviceBuffer(n * 15 * 4); obs0 = multi_gpu.DeviceBuffer(n * 13 * 4); D2 = v.obs_dim + 2
for i in range(WARM):
    if i % 96 == 0: v.reset_device(obs0.ptr)
    v.step_device_packed(acts[i%4].ptr, packed.ptr)
v.sync()
v.profile_begin(STEPS, every=2)
import time
t0 = time.perf_counter()
for i in range(STEPS):
    if i % 96 == 0: v.reset_device(obs0.ptr)
    v.step_device_packed(acts[i%4].ptr, packed.ptr)
a, b, k = v.profile_end()
dt = time.perf_counter() - t0
import numpy as np
# the same steps as hipGraph replays of two episodes (what bench.py times at N = 1): microseconds per step without the host
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
print(os.environ.get("CHUB_LIB","")[-12:], os.environ.get("AB_TILE", ""), os.environ.get("AB_FUSED", ""), os.environ.get("AB_ORDER", ""), n, kw["station_list"], "slot_us %.2f env_us %.2f step_us %.2f graph_us %.2f  checksum %.6f" % (a/k*1e3, b/k*1e3, dt/STEPS*1e6, graph_us, chk))
v.close()
