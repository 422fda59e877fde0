#!/usr/bin/env python3
"""bench.py -- env-steps/s of the charging-hub step path at 65 536 parallel envs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

No PyTorch in here: the host is ctypes + numpy over libchub (the launcher only provides RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT); device buffers, streams, the hipGraph capture and the RCCL gather are libchub's own (include/chub.h).

One "step" = one pass of the hot path (slot kernel + env kernel) over all envs of the workload: 65 536 envs of the
reference test hub [20 fast, 25 slow] (BASELINE.json configs[3]; it fits one GPU, so N=1 runs the same workload).
--scaling strong (default): the 65 536 envs are sharded over the N ranks (global env ids keep the Philox streams identical
for every N); --scaling weak: BASELINE.json configs[4] (hub [32 fast, 32 slow], real price / PV / wind series), 262 144 / 8 =
32 768 envs PER GPU, so the 8-GPU run is the configs[4] job.  Every step each rank's packed (obs, reward, done) block goes
to rank 0 in one RCCL gather (chub_step_gather), stream-ordered behind the step kernels.  Actions are a random policy drawn
on the device before the timed region (8 resident batches, cycled); episodes are reset every 96 steps inside the timed region.

At N = 1 whole episodes are captured into a hipGraph (2 episodes = 2 resets + 192 steps per replay) and the timed region
is graph replays plus, at its end, one segment of 192 steps issued call by call with HIP events around the two kernels
of every 4th step: the per-kernel times behind `roofline` are measured inside the timed region.  At N > 1 every step is a
call by default (16 us of host time per step, below the GPU time of a shard's step + gather); --graph on captures there
too (RCCL inside the capture: verified on a world of one).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the slot kernel) with ALGORITHMIC bytes (DESIGN.md
section 5) over its average duration; `roofline_step` prices the whole step the same way (B * env-steps/s /
8e12).  `cpu_baseline` is the CPU oracle (oracle/chub_oracle.c, kind "port") timed on this box's host cores on a bounded sample.
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
# RCCL between the ranks of a node: this pool's host driver only supports dmabuf IPC (exported in the images; kept for bare environments)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HUB = dict(station_list=[20, 25], station_type_list=["fast", "slow"], constant_charging=False, hydro_prod_rate=100.0,
           hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.0,
           price_fluctuate=0.0, hydro_loss=0.0)
# BASELINE.json configs (c4 = the headline; the others are selectable for the results table)
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
GRAPH_EPISODES = 2     # episodes per captured graph: 2 resets + 192 steps = an even number of launches (double-buffered draws)


def algorithmic_bytes(S, D):
    """B(S, D) = 36*S + 269 + 4*D bytes per env-step; the slot kernel owns the per-slot
    part (16 B state read + 4 B action + 16 B state written per slot) plus the station scalars it hands over
    (2 stations x (line, flow_in, car_number, 3 f32 sums) = 2 x 16 B written, 2 x 1 B line read)."""
    slot_kernel = 36 * S + 34
    env_kernel = (36 * S + 269 + 4 * D) - slot_kernel
    return slot_kernel, env_kernel


def measured_traffic(build_id, envs_per_gpu, total_envs):
    """HBM bytes per slot-kernel launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    gfx950 correction of MI355X_MICROARCH.md), collected offline by tools/refresh_profiles.sh with this same command and
    committed under profiles/ TOGETHER WITH the build id of the library they were measured on: a profile of another build
    is not this build's traffic -> None."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("build_id") == build_id and rec.get("envs") == total_envs:
            return rec["k_slot"]["traffic_bytes_per_launch"] * envs_per_gpu / float(total_envs), os.path.basename(f)
    return None, None


def cpu_baseline(hub_kw, total_envs, target_seconds=12.0):
    """The oracle's scalar restatement (kind "port"), PHILOX streams, all host cores of this box, on a bounded
    sample of the same workload: n_envs chosen so the run takes ~target_seconds; reports env-steps/s."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from orclib import orc, ptr

    cores = min(len(os.sched_getaffinity(0)), 16)  # a one-GPU box's CPU share
    cfg = orclib.make_config(piles=hub_kw["station_list"], types=hub_kw["station_type_list"], hydro_prod_rate=100.0,
                             hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=hub_kw["fcev_permeate"])
    D = orc.orc_env_obs_dim(C.byref(cfg))
    A = sum(hub_kw["station_list"]) + 2

    def run(n, steps):
        h = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 0, orclib.PHILOX, SEED)
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
    n = int(max(64 * cores, min(total_envs, rate * target_seconds / steps)))
    n -= n % cores
    if n == total_envs:  # whole workload fits: add whole episodes until the sample is ~target_seconds long
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
    ap.add_argument("--envs", type=int, default=None, help="total envs (strong) / envs per GPU (weak); default: the config's")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="capture whole episodes into a hipGraph (the last 192 steps of the timed region stay single calls for the HIP "
                         "events); auto: on at N = 1, off at N > 1, where the capture holds RCCL operations of several processes, "
                         "which this repository could only exercise on a world of one")
    ap.add_argument("--force-comm", action="store_true", help="N = 1: still make the communicator and gather (to rank 0 itself)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP events in the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with one process per GPU, e.g. python -m torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    import numpy as np

    import charginghub_env_amd as chub
    from charginghub_env_amd import multi_gpu

    lib = chub.load_library()
    n_dev = lib.chub_device_count()
    if n_dev <= 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank %= n_dev  # a launcher that shows every rank only its own GPU leaves one device, ordinal 0

    config = args.config or ("c5" if args.scaling == "weak" else "c4")
    cfg_envs, hub_kw = CONFIGS[config]
    if args.scaling == "weak":
        per = args.envs if args.envs is not None else cfg_envs // 8  # configs[4]: 262 144 envs over the 8 GPUs of a node
        total = per * world
    else:
        total = args.envs if args.envs is not None else cfg_envs
        assert total % world == 0
        per = total // world
    use_comm = world > 1 or args.force_comm
    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1)
    steps, warmup = args.steps, args.warmup  # exactly W untimed and K timed steps, whatever the launch form
    per_graph = 96 * GRAPH_EPISODES
    # the end of the timed region is issued call by call, with HIP events on every PROFILE_EVERY-th step (all of it when no
    # graph fits in front: a replay starts at an episode boundary and covers per_graph steps)
    eager_tail = 0 if args.no_events else min(steps, per_graph)

    comm = multi_gpu.Comm(rank, world, local_rank) if use_comm else None
    v = chub.VecChargingHub(per, seed=SEED, rng="philox", device=local_rank, env_id0=rank * per, **hub_kw)
    D, A, S = v.obs_dim, v.act_dim, v.n_slots
    stream = multi_gpu.Stream(local_rank)
    row = (D + 2) * 4
    actions = [multi_gpu.DeviceBuffer(per * A * 4, local_rank) for _ in range(N_ACTION_BATCHES)]
    for b, a in enumerate(actions):
        v.random_actions_device(a.ptr, ACTION_KEY, b, stream.ptr)
    # packed step output (obs, reward, done), double-buffered so that a consumer of step i's block is not overwritten by step i+1
    packed = [multi_gpu.DeviceBuffer(per * row, local_rank) for _ in range(2)]
    gathered = [multi_gpu.DeviceBuffer(total * row, local_rank) if (use_comm and rank == 0) else None for _ in range(2)]
    reset_obs = multi_gpu.DeviceBuffer(per * D * 4, local_rank)

    def one_step(i):
        b = i & 1
        if i % 96 == 0:
            v.reset_device(reset_obs.ptr, stream=stream.ptr)
        if use_comm:  # step kernels + ONE grouped ncclSend / ncclRecv on the same stream (chub_step_gather)
            chub._lib.check(lib.chub_step_gather(v._h, comm._h, actions[i % N_ACTION_BATCHES].ptr, packed[b].ptr,
                                                 gathered[b].ptr if rank == 0 else None, stream.ptr))
        else:
            v.step_device_packed(actions[i % N_ACTION_BATCHES].ptr, packed[b].ptr, stream=stream.ptr)

    def fence():
        stream.sync()
        if comm is not None:
            comm.barrier(stream.ptr)
        stream.sync()

    graph = None
    if use_graph:
        stream.sync()
        v.graph_begin(stream.ptr)
        for i in range(per_graph):
            one_step(i)
        graph = v.graph_end(stream.ptr)

    replayed = [0]

    def run(first, n_steps, reserve):
        """steps first .. first + n_steps - 1: graph replays wherever one fits (it starts at an episode boundary and must end
        `reserve` steps before the end of the span), single calls otherwise"""
        i, end = first, first + n_steps
        while i < end:
            if graph is not None and i % 96 == 0 and end - reserve - i >= per_graph:
                v.graph_launch(graph, stream.ptr)
                i += per_graph
                replayed[0] += per_graph
            else:
                one_step(i)
                i += 1
        return i

    run(0, warmup, 0)
    fence()
    replayed[0] = 0
    use_events = not args.no_events
    t0 = time.perf_counter()
    head = steps - eager_tail
    run(warmup, head, 0)
    if use_events:
        v.profile_begin(eager_tail, every=PROFILE_EVERY)
    run(warmup + head, eager_tail, eager_tail)
    t_issue = time.perf_counter() - t0  # host time to issue the whole timed region
    fence()
    dt = time.perf_counter() - t0
    slot_ms = env_ms = 0.0
    n_prof = 0
    if use_events:
        slot_ms, env_ms, n_prof = v.profile_end()
    if comm is not None:
        dt = comm.max(dt, stream.ptr)
    # sanity on the last outputs (rank-local): finite, done flag consistent with the clock
    last = packed[(steps - 1) & 1].to_host(np.float32, (per, D + 2), stream.ptr)
    assert np.isfinite(last).all(), "non-finite step output"
    assert (last[:, D + 1] > 0.5).all() == ((steps % 96) == 0), "done flag out of step with the clock"
    if use_comm and rank == 0:  # the gathered block's own shard is the local block
        g = gathered[(steps - 1) & 1].to_host(np.float32, (total, D + 2), stream.ptr)
        assert np.array_equal(g[:per], last), "gathered block differs from the local one"
        assert np.isfinite(g).all()

    if rank == 0:
        value = total * steps / dt
        slot_b, env_b = algorithmic_bytes(S, D)
        build_id = lib.chub_build_id().decode()
        roofline = None
        if n_prof:
            slot_s = slot_ms / 1e3 / n_prof
            achieved = slot_b * per / slot_s / 1e9
            traffic, traffic_src = measured_traffic(build_id, per, total)
            roofline = {"bound": "hbm", "kernel": "k_slot_packed", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": slot_b * per, "avg_launch_us": slot_s * 1e6,
                        "env_kernel_avg_launch_us": env_ms / n_prof * 1e3,
                        "env_kernel_algorithmic_bytes_per_launch": env_b * per}
        step_achieved = (slot_b + env_b) * total / (dt / steps) / 1e9 / world  # per GPU
        out = {
            "metric": "env-steps/sec at 65 536 parallel envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d envs x hub [%d fast, %d slow] (BASELINE.json %s), fcev_permeate %g, "
                                   "random policy resident in HBM, reset every 96 steps, Philox streams"
                                   % (total, hub_kw["station_list"][0], hub_kw["station_list"][1],
                                      {"c4": "configs[3]", "c5": "configs[4]"}.get(config, config), hub_kw["fcev_permeate"]),
                       "envs_per_gpu": per, "obs_dim": D, "act_dim": A, "launch": ("hipGraph of %d episodes: %d of the %d timed steps are replays, the others (incl. the last %d, with "
                                  "HIP events) are single calls" % (GRAPH_EPISODES, replayed[0], steps, eager_tail))
                       if graph else "every step a call",
                       "host_issue_ms_per_step": t_issue / steps * 1e3,
                       "collective": "none" if not use_comm else
                       "one grouped ncclSend/ncclRecv (RCCL) of [envs_per_gpu, %d] f32 per step to rank 0, on the step's stream" % (D + 2)},
            "roofline": roofline,
            "roofline_step": {"bound": "hbm", "what": "whole step (slot kernel + env kernel + launch gaps), SURVEY.md 8(d): B * env-steps/s per GPU",
                              "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_achieved / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_env_step": slot_b + env_b},
            "build_id": build_id,
        }
        if not args.no_cpu_baseline and world == 1 and config == "c4":
            out["cpu_baseline"] = cpu_baseline(hub_kw, total)
        print(json.dumps(out))
    if graph is not None:
        v.graph_destroy(graph)
    stream.sync()
    v.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
