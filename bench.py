#!/usr/bin/env python3
"""bench.py -- env-steps/s of the charging-hub step path at 65 536 parallel envs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the hot path (chub_step: slot kernel + env kernel) over all 65 536 envs of the
reference test hub [20 fast, 25 slow] (BASELINE.json configs[3]; it fits one GPU, so N=1 runs the same
workload).  The envs are sharded over the N ranks (strong scaling: 65 536 / N envs per GPU, global env ids keep
the Philox streams identical for every N); every step each rank's packed (obs, reward, done) block is gathered
to rank 0 with one RCCL gather, as the north star specifies.  Actions are a random policy drawn on the device
before the timed region (8 resident batches, cycled); episodes are reset every 96 steps inside the timed region.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the slot kernel) with ALGORITHMIC bytes
 over its average duration measured with HIP events on the launch stream; `cpu_baseline` is
the CPU oracle (oracle/chub_oracle.c, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory instead of host memory: every wave's first scalar loads then stay on the GPU
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

TOTAL_ENVS = 65536
HUB = dict(station_list=[20, 25], station_type_list=["fast", "slow"], constant_charging=False, hydro_prod_rate=100.0,
           hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.0,
           price_fluctuate=0.0, hydro_loss=0.0)
# the other BASELINE.json configs (parity-test sizes; selectable with --config for the results table, not the headline)
CONFIGS = {
    "c2": (4096, dict(HUB, station_list=[16, 0], fcev_permeate=0.0)),
    "c3": (32768, dict(HUB)),
    "c4": (65536, dict(HUB)),
    "c5": (262144, dict(HUB, station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3)),
}
SEED = 12345
ACTION_KEY = 0xC0FFEE
N_ACTION_BATCHES = 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_EVERY = 4      # HIP events around the two kernels on every 4th step of the timed region


def algorithmic_bytes(S, D):
    """B(S, D) = 36*S + 269 + 4*D bytes per env-step; the slot kernel owns the per-slot
    part (16 B state read + 4 B action + 16 B state written per slot) plus the station scalars it hands over
    (2 stations x (line, flow_in, car_number, 3 f32 sums) = 2 x 16 B written, 2 x 1 B line read)."""
    slot_kernel = 36 * S + 34
    env_kernel = (36 * S + 269 + 4 * D) - slot_kernel
    return slot_kernel, env_kernel


def measured_traffic(envs_per_gpu):
    """HBM bytes per k_slot launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    corrected with the calibration kernel of tools/microbench/copy4.hip) -- collected offline with this same command
    and committed under profiles/; scaled to this run's shard size."""
    import glob

    import re

    def version(path):
        m = re.search(r"round(\d+)_v(\d+)_pmc_traffic", path)
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")), key=version)
    if not files:
        return None
    try:
        rec = json.load(open(files[-1]))
        return rec["k_slot"]["traffic_bytes_per_launch"] * envs_per_gpu / float(TOTAL_ENVS)  # measured on the c4 hub
    except Exception:
        return None


def cpu_baseline(target_seconds=12.0):
    """The oracle's scalar restatement (kind "port"), PHILOX streams, all host cores of this box, on a bounded
    sample of the same workload: n_envs chosen so the run takes ~target_seconds; reports env-steps/s."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from orclib import orc, ptr

    cores = min(len(os.sched_getaffinity(0)), 16)  # a one-GPU box's CPU share
    cfg = orclib.make_config(piles=HUB["station_list"], types=HUB["station_type_list"], hydro_prod_rate=100.0,
                             hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)

    def run(n, steps):
        h = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 0, orclib.PHILOX, SEED)
        D, A = 13, 47
        obs = np.zeros((n, D))
        rew = np.zeros(n)
        done = np.zeros(n, dtype=np.uint8)
        rs = np.random.RandomState(0)
        acts = [rs.uniform(-1, 1, size=(n, A)).astype(np.float32) for _ in range(4)]
        orc.orc_vec_reset(h, None, None, ptr(obs))
        t0 = time.perf_counter()
        for t in range(steps):
            if t and t % 96 == 0:
                orc.orc_vec_reset(h, None, None, ptr(obs))
            orc.orc_vec_step(h, ptr(acts[t % 4]), None, ptr(obs), ptr(rew), ptr(done), cores)
        dt = time.perf_counter() - t0
        orc.orc_vec_destroy(h)
        return n * steps / dt

    n = 64 * cores
    rate = run(n, 24)  # calibration burst
    steps = 96
    n = int(max(64 * cores, min(TOTAL_ENVS, rate * target_seconds / steps)))
    n -= n % cores
    if n == TOTAL_ENVS:  # whole workload fits: add whole episodes until the sample is ~target_seconds long
        steps = 96 * max(1, min(4, int(round(rate * target_seconds / (n * 96.0)))))
    rate = run(n, steps)
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps of the same hub (oracle/chub_oracle.c, Philox streams, %d pthreads)"
                      % (n, steps, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4800)
    ap.add_argument("--warmup", type=int, default=960)
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP events in the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    import torch  # first: libchub must share torch's HIP runtime (same soname)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal switch (never used by the driver): CHUB_BENCH_REHEARSE=1 runs all ranks on GPU 0 with the gloo
    # backend (host-staged gather), to exercise the multi-rank orchestration on a one-GPU box.
    rehearse = os.environ.get("CHUB_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL

    import charginghub_env_amd as chub

    cfg_envs, hub_kw = CONFIGS[args.config]
    total = args.envs if args.envs is not None else cfg_envs
    assert total % world == 0
    per = total // world
    v = chub.VecChargingHub(per, seed=SEED, rng="philox", device=local_rank, env_id0=rank * per, **hub_kw)
    D, A, S = v.obs_dim, v.act_dim, v.n_slots
    stream = torch.cuda.current_stream().cuda_stream

    actions = [torch.empty((per, A), dtype=torch.float32, device=dev) for _ in range(N_ACTION_BATCHES)]
    for b, a in enumerate(actions):
        v.random_actions_device(a.data_ptr(), ACTION_KEY, b, stream)
    # packed step output (obs, reward, done); two buffers so that the gather of step i (RCCL's own stream) can
    # overlap with the kernels of step i+1, which write the other buffer
    packed = [torch.empty((per, D + 2), dtype=torch.float32, device=dev) for _ in range(2)]
    reset_obs = torch.empty((per, D), dtype=torch.float32, device=dev)
    gathered = [None, None]
    gdev = "cpu" if rehearse else dev
    if world > 1 and rank == 0:
        gathered = [[torch.empty((per, D + 2), dtype=torch.float32, device=gdev) for _ in range(world)] for _ in range(2)]
    pending = [None, None]

    def one_step(i):
        b = i & 1
        if pending[b] is not None:
            pending[b].wait()  # stream-level: the step that reuses this buffer waits for its previous gather
            pending[b] = None
        if i % 96 == 0:
            v.reset_device(reset_obs.data_ptr(), stream=stream)
        v.step_device_packed(actions[i % N_ACTION_BATCHES].data_ptr(), packed[b].data_ptr(), stream=stream)
        if world > 1:
            src = packed[b].cpu() if rehearse else packed[b]
            pending[b] = dist.gather(src, gather_list=gathered[b], dst=0, async_op=True)

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    use_events = not args.no_events
    if use_events:
        v.profile_begin(args.steps, every=PROFILE_EVERY)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    slot_ms = env_ms = 0.0
    n_prof = 0
    if use_events:
        slot_ms, env_ms, n_prof = v.profile_end()

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # sanity on the last outputs (rank-local): finite, done flag consistent with the clock
    last = packed[(args.warmup + args.steps - 1) & 1].cpu()
    assert bool(torch.isfinite(last).all()), "non-finite step output"

    if rank == 0:
        value = total * args.steps / dt
        slot_b, env_b = algorithmic_bytes(S, D)
        roofline = None
        if n_prof:
            slot_s = slot_ms / 1e3 / n_prof
            achieved = slot_b * per / slot_s / 1e9
            roofline = {"bound": "hbm", "kernel": "k_slot_packed (k_slot for hub shapes it does not cover)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(per) if args.config in ("c3", "c4") else None,
                        "algorithmic_bytes_per_launch": slot_b * per, "avg_launch_us": slot_s * 1e6,
                        "env_kernel_avg_launch_us": env_ms / n_prof * 1e3,
                        "env_kernel_algorithmic_bytes_per_launch": env_b * per}
        out = {
            "metric": "env-steps/sec at 65 536 parallel envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d envs x hub [%d fast, %d slow] (BASELINE.json %s), fcev_permeate %g, "
                                   "random policy resident in HBM, reset every 96 steps, Philox streams"
                                   % (total, hub_kw["station_list"][0], hub_kw["station_list"][1],
                                      "configs[3]" if args.config == "c4" else args.config, hub_kw["fcev_permeate"]),
                       "envs_per_gpu": per, "obs_dim": D, "act_dim": A,
                       "collective": "none" if world == 1 else "one RCCL gather of [envs_per_gpu, %d] f32 per step" % (D + 2)},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1 and args.config == "c4":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    v.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
