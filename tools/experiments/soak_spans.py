#!/usr/bin/env python3
"""60 whole days (5760 steps, a reset per day) on three hubs: every step a launch through the one-launch step, spans with the tails on a wave of their
own / on the last slot wave -- against the two-launch step: a digest over both packed blocks at every day's end + the end state.  Ran equal on the
round's final sources (9 of 9).    python3 tools/experiments/soak_spans.py"""
import ctypes as C, sys, hashlib
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check
def run(n, kw, fused, span, tails, days):
    v = chub.VecChargingHub(n, seed=11, fused_step=fused, span_steps=span, span_tails=tails, **kw)
    D, A = v.obs_dim, v.act_dim
    st = multi_gpu.Stream(0)
    acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(5)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, 99, b, st.ptr)
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
    obs0 = multi_gpu.DeviceBuffer(n * D * 4)
    c_acts = (C.c_void_p * 5)(*[a.ptr for a in acts]); c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
    h = hashlib.blake2b(digest_size=16)
    for d in range(days):
        check(v._lib.chub_run_steps(v._h, None, c_acts, 5, c_packed, None, obs0.ptr, 96 * d, 96, st.ptr))
        h.update(packed[0].to_host(np.float32, (n, D + 2), st.ptr).tobytes()); h.update(packed[1].to_host(np.float32, (n, D + 2), st.ptr).tobytes())
    for x in v.slots(): h.update(np.ascontiguousarray(x).tobytes())
    h.update(np.ascontiguousarray(v.station_scalars()).tobytes())
    v.close(); st.destroy()
    return h.hexdigest()
base = dict(hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0)
for n, kw in ((4096, dict(base, station_list=[16, 0], station_type_list=["fast", "slow"], fcev_permeate=0.0)),
              (2816, dict(base, station_list=[20, 25], station_type_list=["fast", "slow"], fcev_permeate=0.08, renew_fluctuate=0.3, price_fluctuate=0.3)),
              (3000, dict(base, station_list=[24, 9], station_type_list=["slow", "fast"], fcev_permeate=0.05, renew_fluctuate=0.1, price_fluctuate=0.1, hydro_loss=0.001))):
    ref = run(n, kw, "off", "off", "auto", 60)
    for name, f, s, t in (("one launch per step", "on", "off", "auto"), ("spans own wave", "on", "auto", "own_wave"), ("spans last slot wave", "on", "auto", "same_wave")):
        got = run(n, kw, f, s, t, 60)
        print(n, kw["station_list"], name, "==" if got == ref else "DIFFERS", flush=True)
