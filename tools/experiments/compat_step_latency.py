import sys, time, os, numpy as np
sys.path.insert(0, ".")
import charginghub_env_amd as chub
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
env = chub.EvcsspManagerEnv_v6(seed_rand=False, rng="compat", **kw)
p = env._p; lib = env._chub_step; h = env._h
best = []
for rep in range(8):
    env.reset()
    t0 = time.perf_counter()
    for t in range(95): lib(h, p[0], p[1], p[2], p[3], p[4])
    best.append((time.perf_counter() - t0) / 95 * 1e6)
print(os.environ.get("CHUB_LIB", "")[-12:], "bare COMPAT chub_step: median %.1f min %.1f us; state checksum %.9f" % (sorted(best)[4], min(best), float(env._tel.sum())))
env.close()
