#!/usr/bin/env python3
"""Three ways to drive the hub, smallest first (run from the repo root on a machine with an MI355X):

    python examples/quickstart.py

1. the drop-in single env (the reference's class and call pattern, test/env_test.py);
2. the batched host-pointer API (numpy in / numpy out);
3. the device-resident API (torch CUDA tensors in / out, no host copies) -- what an on-device learner should use.
"""
import os
import sys
import time

import numpy as np

try:
    import torch  # before libchub is loaded (part 3 shares torch's HIP runtime); parts 1 and 2 do not need it
except ImportError:
    torch = None

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HUB = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100, hydro_store_vlt=25,
           init_soc=0.2, fc_max_power=100, fcev_permeate=0.01)


def single_env():
    import charginghub_env_amd as chub

    env = chub.make("evcssp_env_cpp:charging-hub-v6", constant_charging=False, seed_rand=False, **HUB)
    env.reset()
    done, ret = False, 0.0
    while not done:
        obs, reward, done, info = env.step(action=None)      # None = every pile on, as in the reference's smoke test
        ret += reward
    print("1. drop-in env: one episode, return %.4f, observation %s" % (ret, np.round(obs, 3)))
    env.close()


def batched_host(n=4096):
    import charginghub_env_amd as chub

    hub = chub.VecChargingHub(n, seed=0, **HUB)
    obs = hub.reset()
    rs = np.random.RandomState(0)
    batches = [rs.uniform(-1, 1, (n, hub.act_dim)).astype(np.float32) for _ in range(8)]   # a policy would produce these
    t0 = time.perf_counter()
    for t in range(96):
        obs, reward, done, _ = hub.step(batches[t % 8])
    dt = time.perf_counter() - t0
    print("2. %d envs through numpy: %.1f M env-steps/s, mean reward %.4f, all done: %s"
          % (n, n * 96 / dt / 1e6, reward.mean(), bool(done.all())))
    # every reference env owns its clock: any subset can start a new day (or move) on its own
    obs = hub.reset()
    for t in range(30):
        obs, reward, done, _ = hub.step(rs.uniform(-1, 1, (n, hub.act_dim)).astype(np.float32))
    obs = hub.reset_envs(np.arange(n) % 2 == 0)              # half of the envs abandon their day
    obs, reward, done, _ = hub.step(rs.uniform(-1, 1, (n, hub.act_dim)).astype(np.float32))
    print("   per-env clocks: slots of day now %s (%d clocks)" % (sorted(set(hub.env_clocks().tolist())), hub.clock_groups))
    hub.close()


def device_resident(n=65536):
    import charginghub_env_amd as chub

    if torch is None:
        print("3. skipped: torch is not installed")
        return

    env = chub.TorchHubVecEnv(n, seed=0, **HUB)
    obs = env.reset()
    actions = torch.rand((n, env.act_dim), device="cuda") * 2 - 1
    for _ in range(96):
        env.step(actions)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 960
    for _ in range(steps):
        obs, reward, done, _ = env.step(actions)             # a policy would map obs -> actions here, on the device
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("3. %d envs on the device: %.0f M env-steps/s, mean reward %.4f" % (n, n * steps / dt / 1e6, float(reward.mean())))
    # a policy that emits on / off decisions hands over one bit per pile (+ the two tail floats) instead of a row of floats
    bits, tail = env.pack_bits(actions)
    t0 = time.perf_counter()
    for _ in range(steps):
        obs, reward, done, _ = env.step_bits(bits, tail)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("   the same decisions as bits: %.0f M env-steps/s" % (n * steps / dt / 1e6))
    env.close()


if __name__ == "__main__":
    single_env()
    batched_host()
    device_resident()
