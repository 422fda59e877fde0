#!/usr/bin/env python3
"""The one-launch forms against the two-launch step on RANDOM hub shapes and batch sizes (a stress run beside the fixed shapes of
tests/test_gpu_parity.py): every step a launch with fused_step off (k_slot_packed + k_env) / on (k_step_tailwave or, hubs of fewer than 8 piles,
k_step_fused), and chub_run_steps's spans with the tails on the last slot wave (k_steps_fused) / on a wave of their own (k_steps_piped, where the
hub has 8 piles or more) -- both packed blocks after every call, slot state and station records, bit for bit.
    python3 tools/experiments/shape_sweep.py [n_shapes] [seed]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
from charginghub_env_amd._lib import check

def run(n, kw, fused, span, tails):
    try:
        v = chub.VecChargingHub(n, seed=5, fused_step=fused, span_steps=span, span_tails=tails, **kw)
    except chub.ChubError as e:
        return None, str(e)
    D, A = v.obs_dim, v.act_dim
    st = multi_gpu.Stream(0)
    acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(3)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, 77, b, st.ptr)
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
    obs0 = multi_gpu.DeviceBuffer(n * D * 4)
    c_acts = (C.c_void_p * 3)(*[a.ptr for a in acts])
    c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
    trace = []
    first = 0
    for count in (3, 1, 50, 43, 20, 2):  # (a reset at 0 and at 96; spans of every length up to the day's rest)
        check(v._lib.chub_run_steps(v._h, None, c_acts, 3, c_packed, None, obs0.ptr, first, count, st.ptr))
        first += count
        trace += [packed[0].to_host(np.float32, (n, D + 2), st.ptr), packed[1].to_host(np.float32, (n, D + 2), st.ptr)]
    trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1)]
    info = (v.uses_fused_step,)
    v.close(); st.destroy()
    for b in acts + packed + [obs0]:
        b.free()
    return trace, info

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(n_shapes):
    S0, S1 = int(rs.randint(0, 65)), int(rs.randint(0, 65))
    if it % 6 == 0: S1 = 0
    if it % 6 == 1: S0 = 0
    if it % 6 == 2: S0, S1 = int(rs.randint(1, 5)), int(rs.randint(0, 4))   # fewer than 8 piles: more than 64 envs per workgroup
    if S0 + S1 == 0: S0 = 8
    types = [["fast", "slow"], ["slow", "fast"], ["fast", "fast"], ["slow", "slow"]][rs.randint(4)]
    n = int(rs.choice([37, 300, 777, 2048, 3000, 5000]))
    kw = dict(station_list=[S0, S1], station_type_list=types, hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=float(rs.choice([0.2, 0.5])),
              fc_max_power=100.0, fcev_permeate=float(rs.choice([0.0, 0.02, 0.08])), renew_fluctuate=float(rs.choice([0.0, 0.3])),
              price_fluctuate=float(rs.choice([0.0, 0.2])))
    ref, info = run(n, kw, "off", "off", "auto")
    if ref is None:
        print(it, [S0, S1], types, n, "refused:", info[:80]); continue
    line = "%2d hub %-9s %-14s n %5d :" % (it, [S0, S1], "/".join(types), n)
    for name, fused, span, tails in (("one launch per step", "on", "off", "auto"), ("spans, last slot wave", "on", "auto", "same_wave"),
                                     ("spans, own wave", "on", "auto", "own_wave"), ("spans of <= 5", "on", 5, "auto")):
        got, inf = run(n, kw, fused, span, tails)
        if got is None:
            line += "  [%s: refused]" % name; continue
        ok = len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(ref, got))
        bad += 0 if ok else 1
        line += "  %s %s" % (name, "==" if ok else "DIFFERS")
    print(line, flush=True)
print("DIFFERENCES:", bad)
sys.exit(1 if bad else 0)
