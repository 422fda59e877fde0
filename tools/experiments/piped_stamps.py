#!/usr/bin/env python3
"""When the tail wave of workgroup 0 of k_steps_piped reaches and leaves the step's four barriers (a build with KFLAGS=-DCHUB_PIPED_STAMPS=1, named by CHUB_LIB):
a long wait = the slot waves are the slower side of that interval; no wait = the tail wave is.  python3 tools/experiments/piped_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check
kw = dict(station_list=[16, 0], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.0)
n = 4096
v = chub.VecChargingHub(n, seed=1, **kw)
D, A = v.obs_dim, v.act_dim
st = multi_gpu.Stream(0)
acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
for b, a in enumerate(acts):
    v.random_actions_device(a.ptr, 123, b, st.ptr)
packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
obs0 = multi_gpu.DeviceBuffer(n * D * 4)
c_acts = (C.c_void_p * 4)(*[a.ptr for a in acts])
c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
lib = C.CDLL(os.environ["CHUB_LIB"])
out = (C.c_ulonglong * 16)()
rows = []
for day in range(6):
    check(v._lib.chub_run_steps(v._h, None, c_acts, 4, c_packed, None, obs0.ptr, 96 * day, 96, st.ptr))
    st.sync()
    assert lib.chub_debug_piped_stamps(out) == 0
    rows.append(np.array(list(out)[:9], dtype=np.int64))
names = ["records + requests -> #1", "wait #1", "park, forecourt -> #2", "wait #2", "rest of first half -> #3", "wait #3", "second half, observation -> #4", "wait #4", "flush"]
r = np.array(rows[1:])
d = np.diff(r, axis=1)
print("s_memtime ticks between the tail wave's stamps, step n - 2 of a 96-step span, mean of 5 spans (100 ticks = 1 us if the counter runs at 100 MHz)")
for k in range(8):
    print("%-34s %8.1f" % (names[k + 1] if k % 2 == 0 and False else ["#1 wait", "records", "#2 wait", "park, forecourt", "#3 wait", "rest of first half, second half, observation", "#4 wait", "flush"][k], d[:, k].mean()))
print("reach #1 -> flushed", (r[:, 8] - r[:, 0]).mean())
