#!/usr/bin/env python3
"""env-steps/s through the HOST-pointer entry point chub_step (numpy in / numpy out: PCIe copies and a sync per step),
for the note in DESIGN.md.  bench.py's `value` is the device-resident rate; this one is never reported as `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import charginghub_env_amd as chub  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
          init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
v = chub.VecChargingHub(n, seed=12345, **kw)
a = np.random.RandomState(0).uniform(-1, 1, (n, v.act_dim)).astype(np.float32)
v.reset()
for _ in range(20):
    v.step(a)
t0 = time.perf_counter()
steps = 192
for i in range(steps):
    if i % 96 == 0:
        v.reset()
    v.step(a)
dt = time.perf_counter() - t0
print("host-pointer path: %d envs, %.3f ms/step, %.1f M env-steps/s" % (n, dt / steps * 1e3, n * steps / dt / 1e6))
p = v.pinned_actions()
p[...] = a
t0 = time.perf_counter()
for i in range(steps):
    if i % 96 == 0:
        v.reset()
    v.step(p)
dt = time.perf_counter() - t0
print("  ... actions written into the pinned buffer: %.3f ms/step, %.1f M env-steps/s; PCIe bound %d B per env-step"
      % (dt / steps * 1e3, n * steps / dt / 1e6, (v.act_dim + v.obs_dim + 2) * 4))

v.close()
v = chub.VecChargingHub(n, seed=12345, copy_outputs=False, **kw)
p = v.pinned_actions()
p[...] = a
v.reset()
t0 = time.perf_counter()
for i in range(steps):
    if i % 96 == 0:
        v.reset()
    v.step(p)
dt = time.perf_counter() - t0
print("  ... copy_outputs=False (the handle's pinned arrays are returned, no copies on the Python side): %.3f ms/step, %.1f M env-steps/s"
      % (dt / steps * 1e3, n * steps / dt / 1e6))

# ---- packed actions (chub_step_bits): one bit per pile + the two tail floats, 16 bytes per env instead of 4 * (S + 2)
bits, tail = v.pinned_bits()
v.pack_actions(a, out=(bits, tail))
v.reset()
for _ in range(20):
    v.step_bits(bits, tail)
t0 = time.perf_counter()
for i in range(steps):
    if i % 96 == 0:
        v.reset()
    v.step_bits(bits, tail)
dt = time.perf_counter() - t0
print("packed-action path (chub_step_bits, bits resident in the handle's pinned staging, copy_outputs=False): %.3f ms/step, "
      "%.1f M env-steps/s; PCIe %d B up + %d B down per env-step"
      % (dt / steps * 1e3, n * steps / dt / 1e6, v.bit_words * 8 + 8, (v.obs_dim + 1) * 4 + 1))
t0 = time.perf_counter()
for i in range(steps):
    if i % 96 == 0:
        v.reset()
    v.pack_actions(a, out=(bits, tail))
    v.step_bits(bits, tail)
dt = time.perf_counter() - t0
print("  ... packing the f32 action rows on the host (numpy) inside the loop: %.3f ms/step, %.1f M env-steps/s"
      % (dt / steps * 1e3, n * steps / dt / 1e6))
v.close()
