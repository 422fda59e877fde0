import os, sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
n = 65536
kw = dict(station_list=[20,25], station_type_list=["fast","slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
v = chub.VecChargingHub(n, seed=1, **kw)
lib = v._lib
acts = [multi_gpu.DeviceBuffer(n * 47 * 4) for _ in range(4)]
for b, a in enumerate(acts): v.random_actions_device(a.ptr, 123, b, 0)
packed = multi_gpu.DeviceBuffer(n * 15 * 4); obs0 = multi_gpu.DeviceBuffer(n * 13 * 4)
st = multi_gpu.DeviceBuffer(256 * 8 * 8)
st.from_host(np.zeros(256 * 8, dtype=np.uint64))
v.reset_device(obs0.ptr)
for i in range(50): v.step_device_packed(acts[i%4].ptr, packed.ptr)
v.sync()
lib.chub_debug_stamps.argtypes = [C.c_void_p]; lib.chub_debug_stamps.restype = None
lib.chub_debug_stamps(st.ptr)
acc = []
for i in range(20):
    v.step_device_packed(acts[i%4].ptr, packed.ptr)
    v.sync()
    s = st.to_host(np.uint64, (256, 8)).astype(np.int64)
    acc.append(s)
lib.chub_debug_stamps(None)
a = np.stack(acc)  # [20, 256, 8]
t0 = a[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
rel = a - t0
names = ["entry", "kernel arguments here", "load burst issued", "after barrier 1 (loads landed)", "H2 + money done", "tail done", "flush issued", "stores drained"]
for i, nm in enumerate(names):
    print("%-32s mean %8.1f  min %8.1f  max %8.1f (counter ticks since the first block's entry)" % (nm, rel[:, :, i].mean(), rel[:, :, i].min(), rel[:, :, i].max()))
d = np.diff(a[:, :, :8], axis=2)
print("phase deltas mean:", np.round(d.mean(axis=(0, 1)), 1))
v.profile_begin(64, every=1)
for i in range(64): v.step_device_packed(acts[i%4].ptr, packed.ptr)
x, y, k = v.profile_end()
print("slot_us %.2f env_us %.2f" % (x / k * 1e3, y / k * 1e3))
