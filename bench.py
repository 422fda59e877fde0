#!/usr/bin/env python3
"""bench.py -- env-steps/s of the charging-hub step path at 65 536 parallel envs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Both forms work for N > 1.  Started WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` makes this process the
launcher: it spawns N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT and a private rendezvous directory in
their environment), relays rank 0's JSON line and exits non-zero if any rank does.  The launcher itself never loads libchub
or touches the GPU (no re-exec of a process that has initialised HIP, ever).

No PyTorch in here: the host is ctypes + numpy over libchub; device buffers, streams, the hipGraph capture and the RCCL gather
are libchub's own (include/chub.h).

One "step" = one pass of the hot path (slot kernel + env kernel) over all envs of the workload: 65 536 envs of the
reference test hub [20 fast, 25 slow] (BASELINE.json configs[3]; it fits one GPU, so N=1 runs the same workload).
--scaling strong (default): the 65 536 envs are sharded over the N ranks (global env ids keep the Philox streams identical
for every N); --scaling weak: BASELINE.json configs[4] (hub [32 fast, 32 slow], real price / PV / wind series), 262 144 / 8 =
32 768 envs PER GPU, so the 8-GPU run is the configs[4] job.  Every step each rank's packed (obs, reward, done) block goes
to rank 0 in one RCCL gather (chub_step_gather), stream-ordered behind the step kernels.  Actions are a random policy drawn
on the device before the timed region (8 resident batches, cycled); episodes are reset every 96 steps inside the timed region.

Timed region = EXACTLY the K steps W .. W+K-1, nothing else.  At N = 1 they are hipGraph replays: whole episodes (2 episodes =
2 resets + 192 steps per replay) when K is long, one graph of the K steps themselves when K is short; at N > 1 every step is a
call by default (--graph on captures there too: RCCL inside the capture, verified on a world of one).

The per-kernel times behind `roofline` are NOT taken from the timed window (a short window is a time-of-day sample: the slot
kernel's duration follows the arrival rate of the slot of day): after the timed region the run continues to the next episode
boundary and then steps FIVE WHOLE UNTIMED DAYS with the dispatch's own start / stop timestamps around both kernels of every
fifth step -- every slot of the day exactly once, with the kernels running back to back as in the timed region.
`roofline.avg_launch_us` is that day average -- what `rocprofv3 --kernel-trace --stats` of whole days reports as AverageNs --
at any --steps.  `roofline_c5` does the same for 262 144 envs x hub [32, 32] (configs[4]'s
whole job on one GPU): a working set beyond the 256 MB Infinity Cache, i.e. genuinely HBM-resident.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the slot kernel) with ALGORITHMIC bytes (DESIGN.md
section 5) over its average duration; `roofline_step` prices the whole step the same way (B * env-steps/s /
8e12).  `cpu_baseline` is the CPU oracle (oracle/chub_oracle.c, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
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
GRAPH_EPISODES = 2     # episodes per captured graph: 2 resets + 192 steps = an even number of launches (double-buffered draws)
DAY_AVG_DAYS = 10      # whole days behind `value_day_avg` (the same launch form as the timed region, after it)
SUSTAIN_SECONDS = 2.0  # ... and, where that form is a captured graph, this long of the same replays behind `sustained`: a stretch of GPU time long
                       # enough for an outside clock or a busy-percentage sampler to see (the timed region and the ten days are milliseconds)
SPAN_GRAPH_MAX = 1920  # a timed region of up to this many steps is captured as ONE graph of exactly those steps


def algorithmic_bytes(S, D):
    """B(S, D) = 36*S + 269 + 4*D bytes per env-step; the slot kernel owns the per-slot
    part (16 B state read + 4 B action + 16 B state written per slot) plus the station scalars it hands over
    (2 stations x (line, flow_in, car_number, 3 f32 sums) = 2 x 16 B written, 2 x 1 B line read)."""
    slot_kernel = 36 * S + 34
    env_kernel = (36 * S + 269 + 4 * D) - slot_kernel
    return slot_kernel, env_kernel


def measured_traffic(build_id, envs_per_gpu, total_envs, hub):
    """HBM bytes per slot-kernel launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    gfx950 correction of MI355X_MICROARCH.md), collected offline by tools/refresh_profiles.sh with this same command and
    committed under profiles/ TOGETHER WITH the build id of the library they were measured on: a profile of another build
    (or another workload) is not this build's traffic -> None."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic*.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("build_id") == build_id and rec.get("envs") == total_envs and rec.get("hub", [20, 25]) == list(hub):
            return rec["k_slot"]["traffic_bytes_per_launch"] * envs_per_gpu / float(total_envs), os.path.basename(f)
    return None, None


def cpu_baseline(hub_kw, total_envs, target_seconds=12.0):
    """The oracle's scalar restatement (kind "port") on the reference's own streams (COMPAT: per env the glibc rand() ring and minstd_rand0 with
    libstdc++'s polar normals -- SURVEY.md 8(d): "compat-RNG mode"; the exogenous normals / days numpy would draw come from a seeded
    RandomState, as the reference's host does), pthreads over the host cores this box's share allows, on a bounded sample of the same
    workload: n_envs chosen so the run takes ~target_seconds; reports env-steps/s."""
    import ctypes as C

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from orclib import orc, ptr

    host_cores = len(os.sched_getaffinity(0))
    cores = min(host_cores, 16)  # threads used: a one-GPU box's CPU share is 16 (the pool's process guard sizes worker pools to it)
    cfg = orclib.make_config(piles=hub_kw["station_list"], types=hub_kw["station_type_list"], hydro_prod_rate=100.0,
                             hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=hub_kw["fcev_permeate"])
    D = orc.orc_env_obs_dim(C.byref(cfg))
    A = sum(hub_kw["station_list"]) + 2

    def run(n, steps):
        h = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 0, orclib.COMPAT, SEED)
        obs = np.zeros((n, D))
        rew = np.zeros(n)
        done = np.zeros(n, dtype=np.uint8)
        rs = np.random.RandomState(0)
        acts = [rs.uniform(-1, 1, size=(n, A)).astype(np.float32) for _ in range(4)]
        zs = [np.ascontiguousarray(rs.normal(size=(n, 3))) for _ in range(4)]
        days = np.ascontiguousarray(np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32))
        orc.orc_vec_reset(h, ptr(days), ptr(zs[0]), ptr(obs))
        t0 = time.perf_counter()
        for t in range(steps):
            if t and t % 96 == 0:
                orc.orc_vec_reset(h, ptr(days), ptr(zs[0]), ptr(obs))
            orc.orc_vec_step(h, ptr(acts[t % 4]), ptr(zs[(t + 1) % 4]), ptr(obs), ptr(rew), ptr(done), cores)
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
    out = {"value": rate, "unit": "env-steps/s", "cores": cores, "host_cores": host_cores, "kind": "port",
           "sample": "%d envs x %d steps of the same hub (oracle/chub_oracle.c on the reference's own streams -- COMPAT: glibc rand() + minstd_rand0 "
                     "per env, SURVEY 8(d); %d pthreads = min(the %d cores this process may run on, the 16 of a one-GPU box's CPU share: the "
                     "pool's process guard sizes worker pools to it))" % (n, steps, cores, host_cores)}
    ref = reference_stations_rate(hub_kw)
    if ref:
        out["reference_stations"] = ref
    return out


def reference_stations_rate(hub_kw, target_seconds=3.0):
    """The reference's OWN native core timed beside the port, where its build travelled with the snapshot (oracle/_ref/libchs_ref.so =
    the unmodified CHS.hpp behind oracle/ref_driver.cpp): the hub's two stations stepped with random on / off rows and reset every 96
    steps (Fast/SlowChargeStation::evs_step / evs_reset, CHS.hpp:1188-1231 / 1499-1542) on ONE core -- the reference is single-threaded
    with process-global RNG state.  Stations only: the Python half of the reference env (about half of its 220 us step)
    cannot travel.  None where the library is absent."""
    import ctypes as C

    import numpy as np

    import orclib

    if not orclib.ref_available():
        return None
    try:
        ref = orclib.ref()
    except (OSError, AssertionError):
        return None
    piles, types = hub_kw["station_list"], hub_kw["station_type_list"]
    ref.ref_seed(1, 1)
    st = [ref.ref_station_new(0 if types[k] == "fast" else 1, int(piles[k]), 1, 0) for k in range(2) if piles[k] > 0]
    rs = np.random.RandomState(0)
    rows = [[np.ascontiguousarray(rs.choice([0.0, 1.0], size=int(p)).astype(np.float32)) for p in piles if p > 0] for _ in range(16)]
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:
        for h in st:
            ref.ref_station_reset(h)
        for t in range(96):
            r = rows[t % 16]
            for i, h in enumerate(st):
                ref.ref_station_step(h, r[i].ctypes.data_as(C.c_void_p), r[i].size)
        steps += 96
    dt = time.perf_counter() - t0
    for h in st:
        ref.ref_station_free(h)
    return {"value": steps / dt, "unit": "hub-steps/s", "cores": 1, "kind": "reference",
            "sample": "%d steps of the hub's two stations alone (the unmodified CHS.hpp, oracle/_ref), random on / off rows, reset every 96 "
                      "steps; the Python half of the reference env is not in it" % steps}


# ------------------------------------------------------------------------------------------------ the launcher (N > 1, no WORLD_SIZE)
def launch_ranks(args):
    """`python bench.py --gpus N` by itself: this process becomes the launcher of N fresh rank processes.  It imports
    neither numpy nor the package, never maps libchub / the HIP runtime, and only relays: rank 0's stdout (the JSON line) to
    its own stdout, everything else to stderr.  Exit status: 0 iff every rank exited 0."""
    import shutil
    import socket
    import subprocess
    import tempfile

    n = args.gpus
    with socket.socket() as s:  # a free port: tells concurrent launches apart for anything that keys on MASTER_PORT
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdv = tempfile.mkdtemp(prefix="chub_launch_")  # mode 0700, fresh per launch: the RCCL id travels through it
    procs, outs = [], []
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), CHUB_RENDEZVOUS_DIR=rdv, CHUB_LAUNCHER_PID=str(os.getpid()))
            out = open(os.path.join(rdv, "rank%d.out" % r), "w+")
            outs.append(out)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out,
                                          stderr=None, stdin=subprocess.DEVNULL))
        deadline = time.time() + args.launch_timeout
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py launcher: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                    for q in alive:
                        procs[q].terminate()  # exactly the PIDs this launcher started
            if alive and time.time() > deadline:
                rc = rc or 124
                sys.stderr.write("bench.py launcher: ranks %s still running after %.0f s; stopping them\n" % (sorted(alive), args.launch_timeout))
                for q in alive:
                    procs[q].kill()
                for q in alive:
                    procs[q].wait()
                alive = set()
            if alive:
                time.sleep(0.05)
        lines = []
        for r, out in enumerate(outs):
            out.seek(0)
            text = out.read()
            if r == 0:
                lines = [ln for ln in text.splitlines() if ln.strip()]
            elif text.strip() and not args.dry_run:
                sys.stderr.write(text)
        if args.dry_run:
            children = []
            for r, out in enumerate(outs):
                out.seek(0)
                for ln in out.read().splitlines():
                    if ln.startswith("{"):
                        children.append(json.loads(ln))
            maps = open("/proc/self/maps").read()
            print(json.dumps({"dry_run": True, "launcher_pid": os.getpid(), "children": children, "rendezvous_dir": rdv,
                              "launcher_maps_libchub": "libchub" in maps, "launcher_maps_hip": "libamdhip64" in maps,
                              "exit": rc}))
        else:
            for ln in lines:
                print(ln)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for out in outs:
            out.close()
        shutil.rmtree(rdv, ignore_errors=True)
    sys.stdout.flush()
    raise SystemExit(rc)


PROFILE_DAYS = 5  # whole days stepped for the per-kernel timestamps, every PROFILE_DAYS-th step sampled


def profiled_days(v, span, first_step):
    """PROFILE_DAYS whole days (reset + 96 steps each) issued from `first_step` (an episode boundary) with the dispatch's own start /
    stop timestamps around both kernels of every PROFILE_DAYS-th step: 5 and 96 are coprime, so the 96 samples are every slot of
    the day exactly once -- a day average -- while four steps in five go out as plain launches, so the kernels run back to back
    as they do in the timed region (a timestamp pair on EVERY step slows the host below the GPU's pace, and kernels that start on
    an idle chip measured 3 % shorter than the same kernels in a replayed graph).
    -> (slot kernel us, env kernel us) averaged, steps sampled, host wall seconds per step"""
    assert first_step % 96 == 0
    n = 96 * PROFILE_DAYS
    v.profile_begin(96, every=PROFILE_DAYS)
    t0 = time.perf_counter()
    span(first_step, n)
    slot_ms, env_ms, n_prof = v.profile_end()  # synchronises
    wall = time.perf_counter() - t0
    assert n_prof == 96
    return slot_ms / n_prof * 1e3, env_ms / n_prof * 1e3, n_prof, wall / n


def layout_bytes(S):
    """What THIS build's layout must move per env and slot-kernel launch: per charger slot 4 B of state read + 4 B
    written + 4 B of action row; per (station, env) unit the 16-byte record written + the 4 decoded-draw bytes read; per env the two
    tail actions handed over (8 B).  The class-table rows are read-only tables (L2-resident), excluded like SURVEY 8(d)'s tables."""
    return 12 * S + 2 * (16 + 4) + 8


def roofline_block(slot_us, env_us, slot_b, env_b, per, traffic, traffic_src, n_prof, window, S=None, cache_resident=True):
    achieved = slot_b * per / (slot_us * 1e-6) / 1e9
    lay = layout_bytes(S) * per if S is not None else None
    # `bound` keeps the contract's vocabulary: the roofline the kernel is priced against (`peak`).  `limited_by`: what the counters say
    # limits it -- at the cache-resident headline size VALU issue and memory latency share the time (SQ_WAIT_ANY 50 %
    # of the wave cycles, the streams come from the Infinity Cache); beyond the 256 MB Infinity Cache the streams come from HBM
    return {"bound": "hbm", "limited_by": "latency/issue (cache-resident streams)" if cache_resident else "hbm",
            "kernel": "k_slot_packed", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            # the layout's own compulsory bytes per launch and the fraction of the roofline they come to: how much is left to gain
            # by moving bytes faster (the contract's `frac` prices SURVEY 8(d)'s 36 B per slot, which this layout no longer moves)
            "layout_bytes_per_launch": lay, "frac_of_layout": (lay / (slot_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if lay else None,
            # the bytes the counters saw move per launch over the same duration (L2's memory side; Infinity-Cache hits included):
            # what the kernel really streams, next to the algorithmic figure the fraction is made of
            "traffic_GBps": (traffic / (slot_us * 1e-6) / 1e9) if traffic else None,
            # ... and the three readings side by side: `frac` (SURVEY 8(d)'s bytes), `frac_of_traffic` (the bytes the counters saw), `frac_of_layout`
            "frac_of_traffic": (traffic / (slot_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": slot_b * per, "avg_launch_us": slot_us, "launches_sampled": n_prof,
            "window": window, "env_kernel_avg_launch_us": env_us, "env_kernel_algorithmic_bytes_per_launch": env_b * per,
            "env_kernel_frac": env_b * per / (env_us * 1e-6) / 1e9 / HBM_PEAK_GBS}


def c5_roofline(chub, multi_gpu, lib, device, build_id):
    """The working set that does not fit the Infinity Cache: 262 144 envs x hub [32 fast, 32 slow] (BASELINE.json configs[4],
    the whole job on one GPU).  One warm-up day, then one profiled day as for the headline."""
    import numpy as np

    n, kw = CONFIGS["c5"]
    v = chub.VecChargingHub(n, seed=SEED, rng="philox", device=device, **kw)
    D, A, S = v.obs_dim, v.act_dim, v.n_slots
    stream = multi_gpu.Stream(device)
    acts = [multi_gpu.DeviceBuffer(n * A * 4, device) for _ in range(2)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, ACTION_KEY, b, stream.ptr)
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4, device) for _ in range(2)]
    reset_obs = multi_gpu.DeviceBuffer(n * D * 4, device)

    import ctypes as C
    c_acts = (C.c_void_p * 2)(acts[0].ptr, acts[1].ptr)
    c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)

    def span(first, count):
        chub._lib.check(lib.chub_run_steps(v._h, None, c_acts, 2, c_packed, None, reset_obs.ptr, first, count, stream.ptr))

    span(0, 96)
    stream.sync()
    slot_us, env_us, n_prof, wall_per_step = profiled_days(v, span, 96)
    last = packed[1].to_host(np.float32, (n, D + 2), stream.ptr)
    assert np.isfinite(last).all() and (last[:, D + 1] > 0.5).all()
    slot_b, env_b = algorithmic_bytes(S, D)
    traffic, src = measured_traffic(build_id, n, n, kw["station_list"])
    out = roofline_block(slot_us, env_us, slot_b, env_b, n, traffic, src, n_prof,
                         "%d whole days after one warm-up day, every %dth step sampled: each slot of the day once" % (PROFILE_DAYS, PROFILE_DAYS),
                         S=S, cache_resident=False)
    out["workload"] = "%d envs x hub [%d fast, %d slow] (BASELINE.json configs[4] on one GPU), renew / price fluctuate 0.3" % (
        n, kw["station_list"][0], kw["station_list"][1])
    out["state_bytes"] = n * S * 4  # one 32-bit word per charger slot
    out["step_frac_call_by_call"] = (slot_b + env_b) * n / wall_per_step / 1e9 / HBM_PEAK_GBS
    out["ms_per_step_call_by_call"] = wall_per_step * 1e3
    v.close()
    for b in acts + packed + [reset_obs]:
        b.free()
    stream.destroy()
    return out


def packed_actions_block(chub, multi_gpu, lib, device):
    """Secondary: the same step with the pile decisions as ONE BIT PER PILE + the two tail floats (chub_step_bits_device; the packed
    slot kernel reads the bits themselves) instead of the reference's float rows -- all that action_to_real (MGR:384-393) keeps of a
    row.  Same random policy, same results (tests); 8 B per env and step of action input instead of 4 (S + 2).  Never `value`."""
    import ctypes as C
    import numpy as np

    out = {"what": "env-steps/s with one bit per pile + two tail floats as the action input (chub_step_bits_device), hipGraph replays of 2 "
                   "episodes; kernel times = day averages, calls back to back, every 5th step sampled; the float-row form of the same "
                   "workloads is `value` / `roofline` (c4) and `roofline_c5`"}
    for cfg in ("c4", "c5"):
        n, kw = CONFIGS[cfg]
        v = chub.VecChargingHub(n, seed=SEED, rng="philox", device=device, **kw)
        h, A, D, W = v._h, v.act_dim, v.obs_dim, v.bit_words
        st = multi_gpu.Stream(device)
        bits, tails = [], []
        for b in range(4):
            a = multi_gpu.DeviceBuffer(n * A * 4, device)
            v.random_actions_device(a.ptr, ACTION_KEY, b, st.ptr)
            hb, ht = v.pack_actions(a.to_host(np.float32, (n, A), st.ptr))
            a.free()
            db, dt = multi_gpu.DeviceBuffer(n * W * 8, device), multi_gpu.DeviceBuffer(n * 2 * 4, device)
            chub._lib.check(lib.chub_copy_to_device(device, db.ptr, hb.ctypes.data, n * W * 8, st.ptr))
            chub._lib.check(lib.chub_copy_to_device(device, dt.ptr, ht.ctypes.data, n * 2 * 4, st.ptr))
            st.sync()
            bits.append(db)
            tails.append(dt)
        obs, rew, done = multi_gpu.DeviceBuffer(n * D * 4, device), multi_gpu.DeviceBuffer(n * 4, device), multi_gpu.DeviceBuffer(n, device)

        def step(i):
            if i % 96 == 0:
                v.reset_device(obs.ptr, stream=st.ptr)
            chub._lib.check(lib.chub_step_bits_device(h, bits[i % 4].ptr, tails[i % 4].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))

        for i in range(192):
            step(i)
        st.sync()
        v.graph_begin(st.ptr)
        for i in range(192):
            step(i)
        g = v.graph_end(st.ptr)
        v.graph_launch(g, st.ptr)
        st.sync()
        reps = 10 if cfg == "c4" else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            v.graph_launch(g, st.ptr)
        st.sync()
        per_step = (time.perf_counter() - t0) / (reps * 192)
        v.graph_destroy(g)
        for rep in range(2):  # one untimed day, then PROFILE_DAYS days with the dispatch's own timestamps on every 5th step
            if rep == 1:
                v.profile_begin(96, every=PROFILE_DAYS)
            for i in range(96 if rep == 0 else 96 * PROFILE_DAYS):
                step(i)
        slot_ms, env_ms, k = v.profile_end()
        o = obs.to_host(np.float32, (n, D), st.ptr)
        assert np.isfinite(o).all()
        out[cfg] = {"value": n / per_step, "unit": "env-steps/s", "ms_per_step": per_step * 1e3, "envs": n, "hub": kw["station_list"],
                    "slot_kernel_us": slot_ms / k * 1e3, "env_kernel_us": env_ms / k * 1e3, "launches_sampled": k,
                    "action_bytes_per_env_step": 8 * W + 8}
        v.close()
        for buf in bits + tails + [obs, rew, done]:
            buf.free()
        st.destroy()
    return out


class CompatBatch(object):
    """The reference-exact COMPAT mode as a batch, as `roofline_compat` / `compat_mode` run it: n envs of the headline hub, the library's default
    per-env seeds (env i: srand(SEED + 2 i + 1), minstd_rand0(SEED + 2 i + 2)), the reference constructor replayed, device-resident actions
    and exogenous normals (two batches, alternating), whole days = one reset + 96 steps.  `**form`: chub_options of the handle (slot_kernel,
    walk_ahead) -- the bench runs the default; tests/test_gpu_bench.py runs the same days on slot_kernel="wave" for the digest."""

    def __init__(self, chub, multi_gpu, n, device=0, **form):
        import numpy as np

        kw = {k: v for k, v in HUB.items()}
        self.n, self.n_days = n, 0
        self.v = v = chub.VecChargingHub(n, seed=SEED, rng="compat", device=device, **form, **kw)
        v.compat_replay_constructor()
        D, A = v.obs_dim, v.act_dim
        self.st = st = multi_gpu.Stream(device)
        rs = np.random.RandomState(1)
        self.acts, self.zs = [], []
        for b in range(2):
            a = multi_gpu.DeviceBuffer(n * A * 4, device)
            v.random_actions_device(a.ptr, ACTION_KEY, b, st.ptr)
            z = multi_gpu.DeviceBuffer(n * 3 * 8, device)
            z.from_host(rs.normal(size=(n, 3)), st.ptr)
            self.acts.append(a)
            self.zs.append(z)
        self.day_buf = multi_gpu.DeviceBuffer(n * 2 * 4, device)
        self.day_buf.from_host(np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32), st.ptr)
        self.obs, self.rew, self.done = multi_gpu.DeviceBuffer(n * D * 4, device), multi_gpu.DeviceBuffer(n * 4, device), multi_gpu.DeviceBuffer(n, device)

    def days(self, count):
        v, st = self.v, self.st
        for _ in range(count):
            v.reset_device(self.obs.ptr, self.day_buf.ptr, self.zs[0].ptr, stream=st.ptr)
            for t in range(96):
                v.step_device(self.acts[t & 1].ptr, self.obs.ptr, self.rew.ptr, self.done.ptr, d_exo_z=self.zs[t & 1].ptr, stream=st.ptr)
            st.sync()
        self.n_days += count

    def end_state_digest(self):
        import hashlib

        import numpy as np

        v, n = self.v, self.n
        h = hashlib.blake2b(digest_size=16)
        for a in ([self.obs.to_host(np.float32, (n, v.obs_dim), self.st.ptr), self.rew.to_host(np.float32, (n,), self.st.ptr)] + list(v.slots()) +
                  [v.station_scalars(), v.compat_state()]):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    def close(self):
        self.v.close()
        for b in self.acts + self.zs + [self.day_buf, self.obs, self.rew, self.done]:
            b.free()
        self.st.destroy()


def compat_traffic(build_id, n_envs):
    """HBM-side bytes per COMPAT step (both launches) from profiles/round*_pmc_compat.json, when measured on THIS build at this size"""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_compat.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("build_id") == build_id and rec.get("envs") == n_envs and rec.get("traffic_bytes_per_step"):
            return float(rec["traffic_bytes_per_step"]), os.path.basename(f)
    return None, None


def dropin_block(chub, multi_gpu, lib, device):
    """Secondary: the two paths a user of the reference's own class hits.  (i) EvcsspManagerEnv_v6.step() for ONE env (BASELINE.json
    configs[0]: test/env_test.py's loop; the reference's CPU step takes about 220 us) in the reference-exact
    COMPAT mode and with the device generator: microseconds per step() call, wall clock, whole episodes incl. their reset().
    (ii) the reference-exact COMPAT mode as a batch: env-steps/s at 4 096 and 65 536 envs, device-resident inputs, call by call
    (COMPAT takes its exogenous normals from the caller every step, so it is not graph-captured).  Never `value`."""
    import numpy as np

    out = {"what": "us per EvcsspManagerEnv_v6.step() at one env (whole episodes, wall clock, includes the Python host); COMPAT-mode "
                   "(reference-exact streams) env-steps/s through the device-pointer entry points",
           "reference_us_per_step": 220.0, "reference_source": "BASELINE.md section 2 (the reference's CPU step, one core)"}
    kw = {k: v for k, v in HUB.items()}
    for rng in ("compat", "philox"):
        env = chub.EvcsspManagerEnv_v6(seed_rand=False, rng=rng, device=device, seed=SEED, **kw)
        act = np.random.RandomState(0).uniform(-1, 1, size=(8, env.action_space.shape[0])).astype(np.float32)

        def episode():
            env.reset()
            t0 = time.perf_counter()
            for t in range(96):
                _, _, done, _ = env.step(act[t & 7])
            assert done
            return (time.perf_counter() - t0) / 96

        episode()
        per = sorted(episode() for _ in range(10))
        out["%s_us_per_step" % rng] = per[len(per) // 2] * 1e6
        out["%s_us_per_step_min_max" % rng] = [per[0] * 1e6, per[-1] * 1e6]
        env.close()
    rates = {}
    for n in (4096, 65536):
        run = CompatBatch(chub, multi_gpu, n, device)
        v, st = run.v, run.st
        run.days(1)
        t0 = time.perf_counter()
        run.days(2)
        dt = time.perf_counter() - t0
        o = run.obs.to_host(np.float32, (n, v.obs_dim), st.ptr)
        assert np.isfinite(o).all()
        rates[str(n)] = {"value": n * 192 / dt, "unit": "env-steps/s", "ms_per_step": dt / 192 * 1e3}
        if n == 65536:
            # the reference-exact mode against the same roofline: SURVEY 8(d) bytes over the day averages of its two launches (k_slot_walk2 = the
            # slot pass of the step beside the stream walks of the NEXT one; k_env = the step's tails), every 5th step of 5 days: each slot of the day once
            v.profile_begin(96, every=PROFILE_DAYS)
            run.days(PROFILE_DAYS)
            slot_ms, env_ms, k = v.profile_end()
            assert k == 96
            D = v.obs_dim
            slot_b, env_b = algorithmic_bytes(v.n_slots, D)
            slot_us, env_us = slot_ms / k * 1e3, env_ms / k * 1e3
            ach = (slot_b + env_b) * n / ((slot_us + env_us) * 1e-6) / 1e9
            traffic, traffic_src = compat_traffic(lib.chub_build_id().decode(), n)
            out["roofline_compat"] = {
                "what": "the reference-exact COMPAT step (the reference's own glibc rand() / minstd_rand0 streams walked per env, the charge curves "
                        "evaluated in f64 in the reference's order) at the headline size, priced like `roofline_step`",
                "bound": "hbm", "limited_by": "instruction issue: the slot pass (two slots per lane, the f64 curve work of both packed into one pass per wave) and, "
                                              "in the same launch, the serial stream walks of the next step (one per lane; they make the new cars too); the tails are a latency chain",
                "kernels": "k_slot_walk2 (the slot pass of both stations beside the NEXT step's stream walks, which run two steps ahead of the slots they "
                           "draw for; + k_compat_walk in front of it on a day's first step) + k_env (the step's tails)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": (slot_b + env_b) * n, "slot_pass_and_next_walks_us": slot_us, "tails_us": env_us,
                # HBM-side bytes per step from the counters (both launches; profiles/*_pmc_compat.json of THIS build, else null) next to the algorithmic figure
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic": (traffic / ((slot_b + env_b) * n)) if traffic else None,
                "launches_sampled": k, "frac_call_by_call": (slot_b + env_b) * n / (dt / 192) / 1e9 / HBM_PEAK_GBS,
                "workload": "%d envs x hub [20 fast, 25 slow], COMPAT streams, device-resident actions and normals, call by call" % n,
                # what the %d days this block has stepped (1 + 2 + %d) left behind: tests/test_gpu_bench.py runs the same days on the one-kernel-per-
                # station form (the unit's first lane walking the env's streams in the reference's order) and holds this digest to that one
                "end_state_digest": run.end_state_digest(), "days_stepped": run.n_days}
        run.close()
    out["compat_mode"] = rates
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4800)
    ap.add_argument("--warmup", type=int, default=960)
    ap.add_argument("--envs", type=int, default=None, help="total envs (strong) / envs per GPU (weak); default: the config's")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="the timed steps as hipGraph replays; auto: on at N = 1, off at N > 1, where the capture holds RCCL "
                         "operations of several processes, which this repository could only exercise on a world of one")
    ap.add_argument("--overlap-gather", action="store_true",
                    help="with a communicator: the gather of step k on the communicator's own stream beside the kernels of step k + 1 "
                         "(chub_comm_set_overlap; event edges -- free inside a captured graph, two host calls per step otherwise)")
    ap.add_argument("--force-comm", action="store_true", help="N = 1: still make the communicator and gather (to rank 0 itself)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the untimed profiled day (no `roofline` block)")
    ap.add_argument("--no-day-avg", action="store_true", help="skip the untimed ten days behind value_day_avg / ms_per_step_day_avg")
    ap.add_argument("--no-sustained", action="store_true", help="skip the two seconds of graph replays behind `sustained` (counter and trace passes: "
                                                                  "every replayed kernel is a record)")
    ap.add_argument("--no-c5", action="store_true", help="skip the secondary roofline_c5 block")
    ap.add_argument("--no-bits", action="store_true", help="skip the secondary packed_actions block (one bit per pile as the action input)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the secondary dropin_single_env block (the reference-shaped class at one env, COMPAT mode as a batch)")
    ap.add_argument("--fused", choices=["auto", "on", "off"], default="auto",
                    help="the step as ONE launch (k_step_fused); auto: the library's choice (small batches)")
    ap.add_argument("--work-order", choices=["auto", "dispatch"], default="auto",
                    help="the packed kernels' workgroups: auto = XCD-aware where the library chooses it, dispatch = the dispatcher's order (A/B)")
    ap.add_argument("--dry-run", action="store_true",
                    help="stop every rank before libchub is loaded and print who it is (launch plumbing check, runs without a GPU)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="(tests) with --dry-run: this rank exits with status 3")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: seconds before ranks still running are stopped")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if args.dry_run:
        maps = open("/proc/self/maps").read()
        print(json.dumps({"rank": rank, "world": world, "local_rank": local_rank, "pid": os.getpid(), "ppid": os.getppid(),
                          "rendezvous_dir": os.environ.get("CHUB_RENDEZVOUS_DIR"), "maps_libchub": "libchub" in maps,
                          "maps_hip": "libamdhip64" in maps}))
        if rank == args.dry_run_fail_rank:
            raise SystemExit(3)
        if args.dry_run_fail_rank >= 0:
            time.sleep(30)  # the launcher stops the surviving ranks long before this
        return

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

    comm = multi_gpu.Comm(rank, world, local_rank) if use_comm else None
    overlap = bool(comm is not None and args.overlap_gather)
    if overlap:
        comm.set_overlap(True)
    v = chub.VecChargingHub(per, seed=SEED, rng="philox", device=local_rank, env_id0=rank * per, fused_step=args.fused, work_order=args.work_order,
                            **hub_kw)
    D, A, S = v.obs_dim, v.act_dim, v.n_slots
    v_fused = v.uses_fused_step
    stream = multi_gpu.Stream(local_rank)
    row = (D + 2) * 4
    actions = [multi_gpu.DeviceBuffer(per * A * 4, local_rank) for _ in range(N_ACTION_BATCHES)]
    for b, a in enumerate(actions):
        v.random_actions_device(a.ptr, ACTION_KEY, b, stream.ptr)
    # packed step output (obs, reward, done), double-buffered so that a consumer of step i's block is not overwritten by step i+1
    packed = [multi_gpu.DeviceBuffer(per * row, local_rank) for _ in range(2)]
    gathered = [multi_gpu.DeviceBuffer(total * row, local_rank) if (use_comm and rank == 0) else None for _ in range(2)]
    reset_obs = multi_gpu.DeviceBuffer(per * D * 4, local_rank)

    # who is there: RCCL's own count of the communicator, an all-reduce over it, and the GPU every rank sits on
    n_ranks_seen, comm_count, rank_devices = 1, 1, None
    info = np.zeros(4, dtype=np.int32)
    chub._lib.check(lib.chub_device_info(local_rank, info.ctypes.data))
    me = np.array([rank, local_rank, info[0], info[1], info[2], info[3], os.getpid(), 0], dtype=np.int32)
    if comm is not None:
        comm_count = comm.world_seen()
        n_ranks_seen = comm.ranks_seen(stream.ptr)
        d_me = multi_gpu.DeviceBuffer(me.nbytes, local_rank)
        d_all = multi_gpu.DeviceBuffer(me.nbytes * world, local_rank) if rank == 0 else None
        d_me.from_host(me, stream.ptr)
        comm.gather(d_me.ptr, d_all.ptr if d_all else 0, me.nbytes, stream.ptr)
        if rank == 0:
            everyone = d_all.to_host(np.int32, (world, 8), stream.ptr)
        else:
            stream.sync()
    else:
        everyone = me.reshape(1, 8)
    if rank == 0:
        rank_devices = [{"rank": int(r[0]), "device": int(r[1]), "pci": "%04x:%02x:%02x" % (r[2], r[3], r[4]), "cus": int(r[5]),
                         "pid": int(r[6])} for r in everyone]
        if len({d["pci"] for d in rank_devices}) != world and not args.force_comm:
            raise SystemExit("ranks share a GPU: %s" % rank_devices)

    # every step goes out through chub_run_steps: a whole span of steps issued from C (a reset at every day boundary; per step chub_step_device_packed,
    # or -- with a communicator -- chub_step_gather: the step kernels + ONE grouped ncclSend / ncclRecv on the same stream): no trip through Python per step
    import ctypes as C
    PtrArr = C.c_void_p * N_ACTION_BATCHES
    c_actions = PtrArr(*[a.ptr for a in actions])
    c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
    c_gathered = (C.c_void_p * 2)(gathered[0].ptr, gathered[1].ptr) if (use_comm and rank == 0) else None

    def span(first, n):
        if n > 0:
            chub._lib.check(lib.chub_run_steps(v._h, comm._h if use_comm else None, c_actions, N_ACTION_BATCHES, c_packed, c_gathered,
                                               reset_obs.ptr, first, n, stream.ptr))

    def fence():
        if comm is not None:
            comm.join(stream.ptr)  # (overlapped gathers still out on the communicator's stream)
        stream.sync()
        if comm is not None:
            comm.barrier(stream.ptr)
        stream.sync()

    def ticks_of(first, n):  # launches (resets + steps) of steps first .. first + n - 1
        return n + sum(1 for i in range(first, first + n) if i % 96 == 0)

    # long timed regions: a graph of whole episodes, captured before anything has run (the handle is at the start of a day)
    span_graph_steps = 0
    episode_graph = None
    if use_graph and steps > SPAN_GRAPH_MAX:
        stream.sync()
        v.graph_begin(stream.ptr)
        span(0, per_graph)  # (chub_run_steps: the same resets and steps issued from C; a handle on the one-launch step sends whole spans of steps as ONE launch)
        episode_graph = v.graph_end(stream.ptr)

    replayed = [0]

    def run(first, n_steps):
        """steps first .. first + n_steps - 1: episode-graph replays wherever one fits (it starts at an episode boundary), single
        calls otherwise"""
        i, end = first, first + n_steps
        while i < end:
            if episode_graph is not None and i % 96 == 0 and end - i >= per_graph:
                v.graph_launch(episode_graph, stream.ptr)
                i += per_graph
                replayed[0] += per_graph
            else:  # up to the next place a replay fits (or the end): one C call for the whole span
                j = end
                if episode_graph is not None:
                    nxt = ((i + 95) // 96) * 96
                    if nxt > i and end - nxt >= per_graph:
                        j = nxt
                span(i, j - i)
                i = j

    run(0, warmup)
    fence()
    replayed[0] = 0
    # short timed regions: ONE graph of exactly the timed steps, captured here (a capture runs nothing: the handle stays where
    # the warm-up left it) -- minus at most a few last steps, issued as calls, when the launch count would be odd (the
    # state-independent draws are double-buffered by launch parity)
    span_graph = None
    if use_graph and episode_graph is None and steps >= 2:
        n = steps
        while n > 0 and ticks_of(warmup, n) % 2:
            n -= 1
        if n > 0:
            v.graph_begin(stream.ptr)
            span(warmup, n)
            span_graph = v.graph_end(stream.ptr)
            span_graph_steps = n
            fence()
    t0 = time.perf_counter()
    if span_graph is not None:
        v.graph_launch(span_graph, stream.ptr)
        replayed[0] += span_graph_steps
        run(warmup + span_graph_steps, steps - span_graph_steps)
    else:
        run(warmup, steps)
    t_issue = time.perf_counter() - t0  # host time to issue the whole timed region
    fence()
    dt = time.perf_counter() - t0
    dt_local = dt
    if comm is not None:
        dt = comm.max(dt, stream.ptr)
    # sanity on the last outputs (rank-local): finite, done flag consistent with the clock
    last_i = warmup + steps - 1
    if use_comm and rank == 0:  # the root's kernels write its block straight into the gathered buffer (chub_step_gather: in place, no self-send)
        g = gathered[last_i & 1].to_host(np.float32, (total, D + 2), stream.ptr)
        last = g[:per]
    else:
        last = packed[last_i & 1].to_host(np.float32, (per, D + 2), stream.ptr)
    assert np.isfinite(last).all(), "non-finite step output"
    assert (last[:, D + 1] > 0.5).all() == (((warmup + steps) % 96) == 0), "done flag out of step with the clock"
    if use_comm and rank == 0:  # every other shard has arrived too
        assert np.isfinite(g).all()
        assert ((g[:, D + 1] > 0.5) == (((warmup + steps) % 96) == 0)).all(), "a shard's done flags are out of step"
        assert (np.abs(g[:, :D]).sum(axis=1) > 0).all(), "a shard's rows never arrived"

    # ---- untimed: on to the next episode boundary, then whole days with per-kernel timestamps on every slot of the day once
    slot_us = env_us = 0.0
    n_prof = 0
    i = warmup + steps
    if not args.no_events:
        span(i, (-i) % 96)
        i += (-i) % 96
        slot_us, env_us, n_prof, _ = profiled_days(v, span, i)
        i += 96 * PROFILE_DAYS
    fence()
    # ---- untimed by the contract, reported BESIDE `value`: DAY_AVG_DAYS whole days (resets included) in the launch form of the timed region --
    # the driver's short form (--steps 20 --warmup 5) times slots 5..24 of a day, a night-time window in which few cars arrive and the
    # slot kernel runs 3 % under its day average; this figure does not depend on where a window falls
    day_avg = None
    sustained = None
    if not args.no_day_avg:
        span(i, (-i) % 96)
        i += (-i) % 96
        g_days = episode_graph
        if use_graph and g_days is None:
            stream.sync()
            v.graph_begin(stream.ptr)
            span(i, per_graph)
            g_days = v.graph_end(stream.ptr)
        if g_days is not None:
            v.graph_launch(g_days, stream.ptr)  # (untimed: the first replay of a fresh graph uploads it)
            i += per_graph
        fence()
        t1 = time.perf_counter()
        if g_days is not None:
            for _ in range(DAY_AVG_DAYS // GRAPH_EPISODES):
                v.graph_launch(g_days, stream.ptr)
        else:
            span(i, 96 * DAY_AVG_DAYS)
        fence()
        dt_days = time.perf_counter() - t1
        i += 96 * DAY_AVG_DAYS
        if comm is not None:
            dt_days = comm.max(dt_days, stream.ptr)
        day_avg = (dt_days / (96 * DAY_AVG_DAYS), "hipGraph replays of %d episodes" % GRAPH_EPISODES if g_days is not None else "every step a call (issued from C)")
        if g_days is not None and not args.no_sustained and not args.no_events:
            reps = max(1, int(round(SUSTAIN_SECONDS / (dt_days / (DAY_AVG_DAYS // GRAPH_EPISODES)))))
            fence()
            t2 = time.perf_counter()
            for _ in range(reps):
                v.graph_launch(g_days, stream.ptr)
            fence()
            dt_sus = time.perf_counter() - t2
            if comm is not None:
                dt_sus = comm.max(dt_sus, stream.ptr)
            i += per_graph * reps
            sustained = (dt_sus, per_graph * reps)
        if g_days is not None and g_days is not episode_graph:
            v.graph_destroy(g_days)
    # ---- with a communicator: the step taken apart per rank, so that a measured point explains itself -- this rank's kernels (the
    # day averages above), the gather alone (HIP events around 100 of them back to back), the host's issue time per step in the
    # timed region, the rank's own wall clock per step -- gathered to rank 0 through the communicator
    phases = None
    if comm is not None:
        # (a rank that fails in here leaves the others inside a collective that never completes: it says so and exits non-zero at once,
        # and the launcher -- this file's, or torch.distributed.run -- stops the rest instead of waiting for its timeout)
        try:
            # (the gather as chub_step_gather issues it: in place on the root -- its own block is in the gathered buffer already)
            g_us = comm.gather_us(gathered[0].ptr if rank == 0 else packed[0].ptr, gathered[0].ptr if rank == 0 else 0, per * row, stream.ptr, reps=100)
            mine = np.array([rank, slot_us, env_us, g_us, t_issue / steps * 1e6, dt_local / steps * 1e6], dtype=np.float64)
            d_mine = multi_gpu.DeviceBuffer(mine.nbytes, local_rank)
            d_everyone = multi_gpu.DeviceBuffer(mine.nbytes * world, local_rank) if rank == 0 else None
            d_mine.from_host(mine, stream.ptr)
            comm.gather(d_mine.ptr, d_everyone.ptr if d_everyone else 0, mine.nbytes, stream.ptr)
            if rank == 0:
                phases = d_everyone.to_host(np.float64, (world, mine.size), stream.ptr)
                d_everyone.free()
            else:
                stream.sync()
            d_mine.free()
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("bench.py rank %d: the per-phase block failed (%s): exiting 1 so that the launcher stops the other ranks\n" % (rank, exc))
            sys.stderr.flush()
            os._exit(1)

    if rank == 0:
        value = total * steps / dt
        slot_b, env_b = algorithmic_bytes(S, D)
        build_id = lib.chub_build_id().decode()
        roofline = None
        if n_prof and v_fused:  # ONE kernel does the whole step: it is priced with the whole step's bytes
            achieved = (slot_b + env_b) * per / (slot_us * 1e-6) / 1e9
            # (the dispatch timestamps are per launch: the profiled days run every step as a launch of its own, k_step_fused; the timed region and
            # value_day_avg run whole spans of steps per launch where there is no communicator -- `roofline_step` / `day_avg.roofline_step_frac` price those)
            roofline = {"bound": "hbm", "limited_by": "launch + latency", "kernel": "k_step_fused", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                        "algorithmic_bytes_per_launch": (slot_b + env_b) * per, "avg_launch_us": slot_us, "launches_sampled": n_prof,
                        "window": "%d whole untimed days after the timed region, every %dth step sampled: each slot of the day once, kernels back to back" % (PROFILE_DAYS, PROFILE_DAYS),
                        "env_kernel_avg_launch_us": 0.0}
        elif n_prof:
            traffic, traffic_src = measured_traffic(build_id, per, total, hub_kw["station_list"])
            roofline = roofline_block(slot_us, env_us, slot_b, env_b, per, traffic, traffic_src, n_prof,
                                      "%d whole untimed days after the timed region, every %dth step sampled: each slot of the day once, kernels back to back" % (PROFILE_DAYS, PROFILE_DAYS),
                                      S=S, cache_resident=per * S < 10 << 20)  # (the library's own size rule: the second tile from 10 M slots)
        step_achieved = (slot_b + env_b) * total / (dt / steps) / 1e9 / world  # per GPU
        d0, s0 = divmod(warmup, 96)
        d1, s1 = divmod(warmup + steps - 1, 96)
        if span_graph is not None:
            launch = "one hipGraph of the timed steps: %d of %d are its replay, the others single calls" % (replayed[0], steps)
        elif episode_graph is not None:
            launch = "hipGraph of %d episodes: %d of the %d timed steps are replays, the others single calls" % (GRAPH_EPISODES, replayed[0], steps)
        else:
            launch = "every step a call"
        out = {
            "metric": "env-steps/sec at 65 536 parallel envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d envs x hub [%d fast, %d slow] (BASELINE.json %s), fcev_permeate %g, "
                                   "random policy resident in HBM, reset every 96 steps, Philox streams"
                                   % (total, hub_kw["station_list"][0], hub_kw["station_list"][1],
                                      {"c4": "configs[3]", "c5": "configs[4]"}.get(config, config), hub_kw["fcev_permeate"]),
                       "envs_per_gpu": per, "obs_dim": D, "act_dim": A, "launch": launch,
                       "kernels_per_step": ("1 per SPAN of up to 96 steps (k_steps_fused: every workgroup goes from step to step by itself; chub_run_steps)"
                                            if (v_fused and not use_comm) else "1 (k_step_fused)") if v_fused else "2 (k_slot_packed + k_env)",
                       "work_order": ("XCD-aware (tiles, tail and level workgroups in contiguous eighths per XCD)" if v.uses_xcd_order else
                                      "the dispatcher's" + (" (--work-order dispatch)" if args.work_order == "dispatch" else " (the library's choice at this size)")),
                       "window": "timed: slot %d of day %d .. slot %d of day %d; roofline: whole untimed days afterwards" % (s0, d0, s1, d1),
                       "host_issue_ms_per_step": t_issue / steps * 1e3,
                       "graph": ("on" if (span_graph is not None or episode_graph is not None) else
                                 "off (--graph on captures the RCCL gather with the step kernels; verified on a world of one only, so N > 1 "
                                 "issues every step as a call by default)" if use_comm else "off"),
                       "collective": "none" if not use_comm else
                       "one grouped ncclSend/ncclRecv (RCCL) of [envs_per_gpu, %d] f32 per step to rank 0 (whose own block is written in place by its step kernels), %s" % (
                           D + 2, "on the communicator's own stream beside the next step's kernels (--overlap-gather)" if overlap else "on the step's stream")},
            "n_ranks_seen": n_ranks_seen, "rccl_comm_count": comm_count, "ranks": rank_devices,
            "roofline": roofline,
            "roofline_step": {"bound": "hbm", "what": "whole step (slot kernel + env kernel + launch gaps), SURVEY.md 8(d): B * env-steps/s per GPU",
                              "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_achieved / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_env_step": slot_b + env_b},
            "build_id": build_id,
        }
        if day_avg is not None:
            out["value_day_avg"] = total / day_avg[0]
            out["ms_per_step_day_avg"] = day_avg[0] * 1e3
            out["day_avg"] = {"what": "the same job over %d whole days (%d steps + their resets) after the timed region, untimed by the contract: env-steps/s that do "
                                      "not depend on which slots of the day a short timed window covers" % (DAY_AVG_DAYS, 96 * DAY_AVG_DAYS),
                              "days": DAY_AVG_DAYS, "launch": day_avg[1],
                              "roofline_step_frac": (slot_b + env_b) * per / day_avg[0] / 1e9 / HBM_PEAK_GBS}
        if sustained is not None:
            out["sustained"] = {"what": "the same graph replayed back to back for about %.0f s after everything else (untimed by the contract): whole days, resets "
                                        "included -- GPU time an outside clock can see" % SUSTAIN_SECONDS,
                                "seconds": sustained[0], "steps": sustained[1], "value": total * sustained[1] / sustained[0],
                                "ms_per_step": sustained[0] / sustained[1] * 1e3}
        if phases is not None:
            # what the builder expects of this configuration, from its own parts: a step cannot be shorter than the slowest rank's
            # kernels + one gather (GPU side) nor than the slowest rank's host issue time (call by call, the host issues two kernel
            # launches and one grouped ncclSend / ncclRecv per step); the measured ms_per_step reads against the larger of the two
            # (overlapped gathers: the gather of step k runs beside the kernels of step k + 1, a step costs the larger of the two)
            gpu_us = float(np.maximum(phases[:, 1] + phases[:, 2], phases[:, 3]).max()) if overlap else float((phases[:, 1] + phases[:, 2] + phases[:, 3]).max())
            host_us = float(phases[:, 4].max())
            graphed = span_graph is not None or episode_graph is not None
            out["phases"] = {
                "what": "per rank, microseconds: kernels = day averages of the dispatch timestamps (every 5th step of 5 untimed days), gather = "
                        "HIP events around 100 gathers back to back, host_issue = host time per step to issue the timed region, wall = the "
                        "rank's own wall clock per timed step",
                "per_rank": [{"rank": int(r[0]), "slot_kernel_us": r[1], "env_kernel_us": r[2], "gather_us": r[3], "host_issue_us": r[4],
                              "wall_us_per_step": r[5]} for r in phases],
                "expected_ms_per_step": (gpu_us if graphed else max(gpu_us, host_us)) / 1e3,
                "expected_bound": ("gpu (the larger of kernels and gather: overlapped)" if overlap else "gpu (kernels + gather)") if (graphed or gpu_us >= host_us)
                else "host issue (call by call)",
                "measured_ms_per_step": dt / steps * 1e3}
    for g in (episode_graph, span_graph):
        if g is not None:
            v.graph_destroy(g)
    stream.sync()
    v.close()
    for b in actions + packed + [g for g in gathered if g is not None] + [reset_obs]:
        b.free()
    if comm is not None:
        comm.barrier()
        comm.close()
    if rank == 0:
        if world == 1 and config == "c4" and not args.no_c5 and not args.no_events:
            out["roofline_c5"] = c5_roofline(chub, multi_gpu, lib, local_rank, out["build_id"])
        if world == 1 and config == "c4" and not args.no_bits and not args.no_c5 and not args.no_events:
            out["packed_actions"] = packed_actions_block(chub, multi_gpu, lib, local_rank)
        if world == 1 and config == "c4" and not args.no_dropin and not args.no_c5 and not args.no_events:
            out["dropin_single_env"] = dropin_block(chub, multi_gpu, lib, local_rank)
            if "roofline_compat" in out["dropin_single_env"]:
                out["roofline_compat"] = out["dropin_single_env"].pop("roofline_compat")
        if not args.no_cpu_baseline and world == 1 and config == "c4":
            out["cpu_baseline"] = cpu_baseline(hub_kw, total)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
