"""GPU parity tests: the HIP path (through the C ABI of include/chub.h) against
  (a) golden trajectories recorded from the unmodified reference (COMPAT streams), and
  (b) the CPU oracle on the same seeds (PHILOX streams),
plus size-independent properties at BASELINE.json's full sizes.

Bars: integer / index quantities (occupancy, charge flags, stay times, queue lengths, arrivals, done)
bit-exact; per-slot floats (soc, power, emergency, target) bit-exact; station power sums, observation,
reward and telemetry within 1e-5 relative (north star), in practice ~1e-7 (sums are reduced in f64 on the
GPU, sequentially in f32 by the reference).
"""
import ctypes as C

import numpy as np
import pytest

import orclib
from orclib import orc, ptr

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # the north star's bar for floats
TIGHT = 1e-9  # what the f64 tail actually achieves once the station sums agree bit for bit


def hub():
    import charginghub_env_amd as chub
    return chub


def kwargs_of(g):
    return dict(station_list=[int(x) for x in g["kw_station_list"]],
                station_type_list=["fast" if int(x) == 0 else "slow" for x in g["kw_station_type"]],
                constant_charging=bool(g["kw_constant_charging"]), hydro_prod_rate=float(g["kw_hydro_prod_rate"]),
                hydro_store_vlt=float(g["kw_hydro_store_vlt"]), init_soc=float(g["kw_init_soc"]),
                fc_max_power=float(g["kw_fc_max_power"]), fcev_permeate=float(g["kw_fcev_permeate"]),
                renew_fluctuate=float(g["kw_renew_fluctuate"]), price_fluctuate=float(g["kw_price_fluctuate"]),
                hydro_loss=float(g["kw_hydro_loss"]))


def close(a, b, what, rtol=RTOL, atol=1e-6):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    bad = np.nonzero(~np.isclose(a, b, rtol=rtol, atol=atol))
    assert bad[0].size == 0, (what, [tuple(int(x[j]) for x in bad) for j in range(min(5, bad[0].size))],
                              a[bad][:5], b[bad][:5])


def check_slots(got, want, what):
    """got/want: [9, n] f32 -- every field bit-exact"""
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (what, got, want)





@pytest.mark.parametrize("name", orclib.GOLDEN_ENV)
@pytest.mark.parametrize("n_envs,migrate", [(1, False), (5, False), (3, True), (4, False), (6, False)], ids=["1", "5", "3_restored", "4_per_station", "6_own_walks"])
def test_compat_matches_reference_golden(name, n_envs, migrate):
    """COMPAT streams: the GPU reproduces the reference trajectories (every env of the batch is given the
    same seeds / tape, so each must equal the recorded single-env run).  migrate: in the middle of every episode the state is
    taken out (chub_get_state), the handle destroyed, a fresh one created and restored (chub_set_state) -- row (f)2 against the
    reference's own recorded run."""
    chub = hub()
    g = orclib.load_golden(name)
    kw = kwargs_of(g)
    # a handful of COMPAT envs run reset / step as ONE launch (k_compat_small: both station passes and the tail back to back); the
    # batch of 5 is kept on the form of large batches (the split step: empties, stream walks one env per lane, slots), the batch of 4 on
    # one kernel per station with the unit's first lane walking -- every fixture pins all three
    # (round 5: the split step walks step i + 1's streams beside step i's tails, k_env_walk; the batch of 6 keeps every step's walk in its own
    # call, chub_options.walk_ahead = 1)
    kw["fused_step"] = "off" if n_envs in (4, 5, 6) else "auto"
    if n_envs in (4, 5, 6):
        kw["slot_kernel"] = "wave" if n_envs == 4 else "packed"  # (COMPAT handles: one kernel per station / the split step, whatever the size)
    if n_envs == 6:
        kw["walk_ahead"] = "off"
    v = chub.VecChargingHub(n_envs, rng="compat", **kw)
    v.set_telemetry(True)
    S0, S1 = kw["station_list"]
    rep = lambda a: np.repeat(np.asarray(a)[None, :], n_envs, axis=0)
    # the reference's constructor: station constructors' evs_reset + the 101-step electrolyser sweep with live FCEV demand
    # (HYD:154-157), on the stream seeds in force at construction (recorded; env_c1_envtest: the process defaults).  The
    # device replays it: the streams advance as the reference's did and hy_power_speed_list comes out as recorded --
    # nothing is injected, also where a tank clamp binds during the sweep (env_c5_random, env_full_tank).
    v.set_compat_seeds(rep(g["ctor_seeds"]))
    v.compat_replay_constructor()
    for e in range(n_envs):
        close(v.hy_table(env=e), g["hy_table"], (name, "hy_table", e), rtol=1e-13, atol=1e-12)
    v.reset(rep(g["ctor_days"]), rep(g["ctor_z"]))  # constructor's reset (MGR:120): shapes the OU states
    seeds = {int(ep): (int(a), int(b)) for ep, a, b in g["seeds"]}
    steps = int(g["steps_per_episode"])
    i = 0
    for ep in range(int(g["episodes"])):
        if ep in seeds:
            v.set_compat_seeds(rep(seeds[ep]))
        v.reset(rep(g["reset_days"][ep]), rep(g["reset_z"][ep]))
        close(v.obs_f64(), rep(g["reset_obs"][ep]), (name, "reset obs", ep), rtol=TIGHT, atol=TIGHT)
        sc = v.station_scalars()
        for e in range(n_envs):
            got = np.concatenate([sc[e, 0, :6], sc[e, 1, :6]])
            assert np.array_equal(got, g["reset_stations"][ep]), (name, "reset stations", ep, got)
        for t in range(steps):
            if migrate and t == steps // 2 + ep:
                snap = v.get_state()
                v.close()
                v = chub.VecChargingHub(n_envs, rng="compat", **kw)
                v.set_telemetry(True)
                v.set_state(snap)
            obs, rew, done, _ = v.step(rep(g["action"][i]), rep(g["exo_z"][i]))
            sl = v.slots()
            sc = v.station_scalars()
            tel = v.telemetry()
            o64 = v.obs_f64()
            r64 = v.reward_f64()
            for e in range(n_envs):
                check_slots(sl[0][e], g["slots0"][i], (name, ep, t, "station0"))
                check_slots(sl[1][e], g["slots1"][i], (name, ep, t, "station1"))
                got = np.concatenate([sc[e, 0, :6], sc[e, 1, :6]])
                want = g["stations"][i]
                assert np.array_equal(got, want), (name, ep, t, got, want)  # sums in the reference's f32 order
                assert bool(done[e]) == bool(g["done"][i])
                assert np.array_equal(tel[e, 19:22], g["telem"][i][19:22]), (name, ep, t, "fcev ints")
                close(o64[e], g["obs"][i], (name, "obs", ep, t), rtol=TIGHT, atol=TIGHT)
                close(obs[e], g["obs"][i], (name, "obs f32", ep, t), atol=1e-6)
                close(r64[e], g["reward"][i], (name, "reward", ep, t), rtol=TIGHT, atol=TIGHT)
                close(tel[e, :19], g["telem"][i][:19], (name, "telemetry", ep, t), rtol=TIGHT, atol=1e-7)
            i += 1
    v.close()


def _oracle_vec(cfg_kw, n, env_id0, seed, tables=None):
    cfg = orclib.make_config(piles=cfg_kw["station_list"], types=cfg_kw["station_type_list"],
                             **{k: cfg_kw[k] for k in cfg_kw if k not in ("station_list", "station_type_list")})
    h = orc.orc_vec_create(C.byref(cfg), tables or orclib.tables(), n, env_id0, orclib.PHILOX, seed)
    return cfg, h


PHILOX_CASES = [
    ("c2", dict(station_list=[16, 0], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
                init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.0), 256),
    ("c3", dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
                init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01), 192),
    ("c5", dict(station_list=[32, 32], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
                init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.3, price_fluctuate=0.3), 128),
    # a multiple of 2048 envs: tiles, tail AND level workgroups in XCD-aware order (chub_options.work_order, k_env's level branch)
    ("c3_xcd", dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
                    init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.2), 2048),
    ("ragged", dict(station_list=[3, 7], station_type_list=["slow", "fast"], hydro_prod_rate=430.0,
                    hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=50.0, fcev_permeate=0.05, hydro_loss=0.001), 77),
    ("one_pile", dict(station_list=[1, 0], station_type_list=["fast", "fast"], hydro_prod_rate=100.0,
                      hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01), 130),
    ("max64", dict(station_list=[64, 64], station_type_list=["slow", "fast"], hydro_prod_rate=2000.0,
                   hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=100.0, fcev_permeate=0.2,
                   constant_charging=False), 33),
    ("constant", dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0,
                      hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01,
                      constant_charging=True), 64),
    # busy forecourts: the 15-minute FIFO gets stuck in most envs and the waiting list grows to hundreds of cars (the
    # reference's list is unbounded, HYD:264-279)
    ("fcev_stuck", dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0,
                        hydro_store_vlt=400.0, init_soc=0.6, fc_max_power=100.0, fcev_permeate=0.1), 48),
    # more than 64 piles per station (the reference takes any size, CHS.hpp:1148,1458): units span several waves, 64-bit sums
    ("big_100_70", dict(station_list=[100, 70], station_type_list=["fast", "slow"], hydro_prod_rate=2000.0,
                        hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=100.0, fcev_permeate=0.02), 19),
    ("big_256", dict(station_list=[256, 0], station_type_list=["slow", "fast"], hydro_prod_rate=2000.0,
                     hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=100.0, fcev_permeate=0.01), 7),
    # a 3-pile fast station: evs_reset can record a negative flow_in (CHS.hpp:1276, 832-842, 1617)
    ("small_fast", dict(station_list=[3, 2], station_type_list=["fast", "fast"], hydro_prod_rate=100.0,
                        hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01), 200),
]


@pytest.mark.parametrize("label,kw,n", PHILOX_CASES, ids=[c[0] for c in PHILOX_CASES])
def test_philox_matches_oracle(label, kw, n):
    """PHILOX streams: GPU == CPU oracle on identical seeds / actions, 2 episodes + a cut-short one."""
    _philox_parity(label, kw, n)


@pytest.mark.parametrize("label,kernel", [("c3", "wave"), ("constant", "wave"), ("c5", "wave"), ("max64", "wave"), ("c2", "wave"),
                                          ("big_100_70", "wave"), ("big_256", "wave")])
def test_philox_other_slot_kernel(label, kernel):
    """PHILOX steps have two slot kernels (wave-local units / units packed end to end over the workgroup) and the
    library picks the packed one wherever the hub shape allows: force the other one (chub_options.slot_kernel) through
    the same parity check"""
    kw, n = next((c[1], c[2]) for c in PHILOX_CASES if c[0] == label)
    _philox_parity(label + "_" + kernel, kw, n, plan=(96, 30), slot_kernel=kernel)


@pytest.mark.parametrize("label", ["c2", "c3", "c5", "ragged", "max64", "constant", "fcev_stuck", "small_fast"])
@pytest.mark.parametrize("fused", ["on", "off"])
def test_philox_single_launch_step(label, fused):
    """PHILOX lock-step steps have two launch forms -- slot kernel + tail kernel, and ONE launch in which every workgroup also
    runs the tails of its envs and draws their next step (k_step_fused: the default for small batches): force each one
    (chub_options.fused_step) through the same parity check against the oracle"""
    kw, n = next((c[1], c[2]) for c in PHILOX_CASES if c[0] == label)
    _philox_parity(label + "_fused_" + fused, kw, n, plan=(96, 20), fused_step=fused)


def test_single_launch_step_is_bit_identical_and_replays_in_graphs():
    """the two launch forms of the step against each other over whole episodes (packed outputs bit for bit), the fused form also
    as hipGraph replays; a handle too large for the default does not pick it"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    n = 2000
    res = {}
    for form in ("off", "on", "graph"):
        v = chub.VecChargingHub(n, seed=4, fused_step="off" if form == "off" else "on", **kw)
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 77, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
        obs0 = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)

        def steps(first, count):
            for i in range(first, first + count):
                if i % 96 == 0:
                    v.reset_device(obs0.ptr, stream=st.ptr)
                v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)

        trace = []
        if form == "graph":
            st.sync()
            v.graph_begin(st.ptr)
            steps(0, 192)
            g = v.graph_end(st.ptr)
            for _ in range(2):
                v.graph_launch(g, st.ptr)
                trace.append(packed[1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
            v.graph_destroy(g)
        else:
            for rep in range(2):
                steps(0, 192)
                trace.append(packed[1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        steps(0, 7)
        trace.append(packed[0].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1)]
        res[form] = trace
        v.close()
        st.destroy()
    for k in range(len(res["off"])):
        assert np.array_equal(res["off"][k], res["on"][k]), ("fused vs two launches", k)
        assert np.array_equal(res["off"][k], res["graph"][k]), ("fused in a graph vs two launches", k)
    small = chub.VecChargingHub(4096, seed=1, station_list=[16, 0], station_type_list=["fast", "slow"], fcev_permeate=0.0)
    large = chub.VecChargingHub(65536, seed=1, **kw)
    assert small.uses_fused_step and not large.uses_fused_step
    small.close()
    large.close()


@pytest.mark.parametrize("piles", [(20, 25), (32, 32), (3, 0), (0, 7), (100, 70), (256, 1)])
def test_large_tile_is_bit_identical(piles):
    """the packed slot kernel's second workgroup tile (512 lanes x 4 slots: handles whose state streams from HBM, chub_options.tile) against
    the default 256 x 2 on the same handle arguments: resets and steps over whole days, a masked reset and masked steps (per-env clocks),
    steps fed one bit per pile, stations without piles and of more than 64 -- packed outputs, slot state and station records bit for bit"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    kw = dict(station_list=list(piles), station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    n = 1500 if sum(piles) < 128 else 300
    res = {}
    for tile in ("small", "large"):
        v = chub.VecChargingHub(n, seed=6, tile=tile, fused_step="off", **kw)
        lib, h = v._lib, v._h
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 78, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
        obs, rew, done = multi_gpu.DeviceBuffer(n * v.obs_dim * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)
        trace = []
        for i in range(110):
            if i % 96 == 0:
                v.reset_device(obs.ptr, stream=st.ptr)
                trace.append(obs.to_host(np.float32, (n, v.obs_dim), st.ptr))
            v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)
            if i % 13 == 0:
                trace.append(packed[i & 1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        hb, ht = v.pack_actions(acts[1].to_host(np.float32, (n, v.act_dim), st.ptr))  # one bit per pile
        db, dt = multi_gpu.DeviceBuffer(hb.nbytes), multi_gpu.DeviceBuffer(ht.nbytes)
        db.from_host(hb, st.ptr)
        dt.from_host(ht, st.ptr)
        v.step_bits_device(db.ptr, dt.ptr, obs.ptr, rew.ptr, done.ptr, stream=st.ptr)
        trace.append(obs.to_host(np.float32, (n, v.obs_dim), st.ptr))
        rs = np.random.RandomState(3)
        for k in range(6):  # per-env clocks: masked resets and steps
            m = np.ascontiguousarray(rs.uniform(size=n) < 0.4, dtype=np.uint8)
            if k == 1:
                check(lib.chub_reset_envs_device(h, m.ctypes.data, None, None, obs.ptr, st.ptr))
            else:
                check(lib.chub_step_envs_device(h, m.ctypes.data, acts[k % 4].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))
            trace.append(obs.to_host(np.float32, (n, v.obs_dim), st.ptr))
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), v.env_clocks()]
        res[tile] = trace
        v.close()
        st.destroy()
    for k, (a, b) in enumerate(zip(res["small"], res["large"])):
        assert np.array_equal(a, b), ("large tile vs small tile", piles, k)


@pytest.mark.parametrize("n", [4096, 2500, 6144 + 256])
def test_work_order_is_bit_identical(n):
    """chub_options.work_order: the XCD-aware order of the packed kernels' workgroups (tiles, tail and level workgroups in contiguous eighths per
    XCD; the level workgroups only where the launch divides evenly: 4096) against the dispatcher's order on the same handle arguments --
    resets, steps over a day and beyond, steps as graph replays: packed outputs, slot state, station records, clocks bit for bit"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    res = {}
    for order in ("auto", "dispatch"):
        v = chub.VecChargingHub(n, seed=9, work_order=order, fused_step="off", **kw)
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 41, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
        obs = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)
        trace = []
        for i in range(100):
            if i % 96 == 0:
                v.reset_device(obs.ptr, stream=st.ptr)
                trace.append(obs.to_host(np.float32, (n, v.obs_dim), st.ptr))
            v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)
            if i % 9 == 0:
                trace.append(packed[i & 1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        st.sync()
        v.graph_begin(st.ptr)
        for i in range(6):
            v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)
        g = v.graph_end(st.ptr)
        v.graph_launch(g, st.ptr)
        trace.append(packed[1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        v.graph_destroy(g)
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), v.env_clocks()]
        res[order] = trace
        v.close()
        st.destroy()
    for k, (a, b) in enumerate(zip(res["auto"], res["dispatch"])):
        assert np.array_equal(a, b), ("XCD-aware order vs the dispatcher's", n, k)


@pytest.mark.parametrize("piles,types", [((4, 5), ("fast", "slow")), ((7, 13), ("slow", "fast")), ((33, 45), ("fast", "slow")),
                                         ((63, 64), ("slow", "fast")), ((5, 64), ("fast", "fast")), ((21, 4), ("slow", "slow")),
                                         ((10, 6), ("fast", "slow"))])
def test_packed_kernel_shape_sweep(piles, types):
    """the packed slot kernel lays whole units end to end over the workgroup, so where a unit meets a wave boundary
    depends on the pile count: sweep awkward counts (and an env count that fills no workgroup evenly)"""
    kw = dict(station_list=list(piles), station_type_list=list(types), hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.03)
    _philox_parity("sweep_%d_%d" % piles, kw, 37, plan=(40,))


def test_philox_user_series(tmp_path):
    """user-supplied arrival CDFs / price / PV / wind through a data directory: same parity bar"""
    from charginghub_env_amd import data_io
    rs = np.random.RandomState(11)
    rates = 45 + 35 * np.sin(np.arange(96) * 2 * np.pi / 96 + 1.0)
    price = 0.12 + 0.08 * np.sin(np.arange(96) * 2 * np.pi / 96) + rs.uniform(0, 0.02, 96)
    pv = np.maximum(0, 30 * np.sin((np.arange(96) - 24) * np.pi / 48))[None, :] * rs.uniform(0.2, 1.2, (9, 1))
    wd = rs.uniform(0, 80, (150, 96))
    d = data_io.write_data_dir(str(tmp_path / "hub"), arrival_cdf=data_io.cdf_from_rates(rates), price=price, pv=pv, wd=wd)
    tables = orc.orc_tables_load(d.encode())
    assert tables
    kw = dict(station_list=[12, 20], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    _philox_parity("user_series", kw, 96, data_dir=d, tables=tables, plan=[96, 20])


def _philox_parity(label, kw, n, data_dir=None, tables=None, plan=(96, 96, 10, 30), slot_kernel="auto", fused_step="auto"):
    chub = hub()
    kw = dict(kw)
    for k, d in (("constant_charging", False), ("renew_fluctuate", 0.0), ("price_fluctuate", 0.0), ("hydro_loss", 0.0)):
        kw.setdefault(k, d)
    seed, env_id0 = 0xC0FFEE12345, 1000
    v = chub.VecChargingHub(n, seed=seed, rng="philox", env_id0=env_id0, data_dir=data_dir, slot_kernel=slot_kernel, fused_step=fused_step, **kw)
    if fused_step != "auto":
        assert v.uses_fused_step == (fused_step == "on")
    v.set_telemetry(True)
    cfg, h = _oracle_vec(kw, n, env_id0, seed, tables)
    D, A = v.obs_dim, v.act_dim
    S0, S1 = kw["station_list"]
    rs = np.random.RandomState(7)
    o_obs = np.zeros((n, D))
    o_rew = np.zeros(n)
    o_done = np.zeros(n, dtype=np.uint8)
    for ep, steps in enumerate(plan):
        g_obs = v.reset()
        orc.orc_vec_reset(h, None, None, ptr(o_obs))
        close(v.obs_f64(), o_obs, (label, "reset obs", ep), rtol=TIGHT, atol=TIGHT)
        sc = v.station_scalars()
        for e in range(n):  # the station records right after evs_reset, incl. a negative flow_in of a small fast station (CHS.hpp:1617)
            for k in (0, 1):
                ws = np.zeros(8)
                orc.orc_station_scalars(orc.orc_env_station(orc.orc_vec_env(h, e), k), ptr(ws))
                assert np.array_equal(sc[e, k, :6], ws[:6]), (label, "reset stations", ep, e, k, sc[e, k], ws)
        for t in range(steps):
            act = rs.uniform(-1, 1, size=(n, A)).astype(np.float32)
            if t % 7 == 0:
                act[:, :S0 + S1] = 1.0
            obs, rew, done, _ = v.step(act)
            orc.orc_vec_step(h, ptr(act), None, ptr(o_obs), ptr(o_rew), ptr(o_done), 4)
            sl = v.slots()
            sc = v.station_scalars()
            tel = v.telemetry()
            for e in range(n):
                env = orc.orc_vec_env(h, e)
                for k, nk in ((0, S0), (1, S1)):
                    want = np.zeros((9, nk), dtype=np.float32)
                    orc.orc_station_slots(orc.orc_env_station(env, k), ptr(want))
                    check_slots(sl[k][e], want, (label, ep, t, e, k))
                    ws = np.zeros(8)
                    orc.orc_station_scalars(orc.orc_env_station(env, k), ptr(ws))
                    assert np.array_equal(sc[e, k, :6], ws[:6]), (label, ep, t, e, k, sc[e, k], ws)
                wt = np.zeros(38)
                orc.orc_env_telemetry(env, ptr(wt))
                assert np.array_equal(tel[e, 19:24], wt[19:24]), (label, ep, t, e, tel[e, 19:24], wt[19:24])
                close(tel[e, :19], wt[:19], (label, "telemetry", ep, t, e), rtol=TIGHT, atol=1e-7)
                close(tel[e, 24:28], wt[24:28], (label, "telemetry (after the fuel cell)", ep, t, e), rtol=TIGHT, atol=1e-7)
                assert np.array_equal(tel[e, 28:38], wt[28:38]), (label, "telemetry (station scalars)", ep, t, e, tel[e, 28:38], wt[28:38])
                assert orc.orc_env_q_overflow(env) == 0
            assert np.array_equal(done, o_done.astype(bool))
            close(v.obs_f64(), o_obs, (label, "obs", ep, t), rtol=TIGHT, atol=TIGHT)
            close(obs, o_obs, (label, "obs f32", ep, t), atol=1e-6)
            close(v.reward_f64(), o_rew, (label, "reward", ep, t), rtol=TIGHT, atol=TIGHT)
    orc.orc_vec_destroy(h)
    v.close()


def test_full_size_properties():
    """65 536 envs (BASELINE.json configs[3] per-node size): determinism, shard independence (Philox counters use
    global env ids), invariants the reference asserts (MGR:191,205; HYD:111-112), occupancy bookkeeping."""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    N = 65536
    whole = chub.VecChargingHub(N, seed=12345, rng="philox", **kw)
    a = chub.VecChargingHub(N // 2, seed=12345, rng="philox", env_id0=0, **kw)
    b = chub.VecChargingHub(N // 2, seed=12345, rng="philox", env_id0=N // 2, **kw)
    rs = np.random.RandomState(3)
    o = whole.reset()
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(o, np.concatenate([oa, ob]))
    ret = np.zeros(N)
    for t in range(96):
        act = rs.uniform(-1, 1, size=(N, whole.act_dim)).astype(np.float32)
        o, r, d, _ = whole.step(act)
        oa, ra, da, _ = a.step(act[:N // 2])
        ob, rb, db, _ = b.step(act[N // 2:])
        assert np.array_equal(o, np.concatenate([oa, ob])) and np.array_equal(r, np.concatenate([ra, rb]))
        assert np.all(np.isfinite(o)) and np.all(np.isfinite(r))
        assert d.all() == (t == 95) and d.any() == (t == 95)
        ret += r
        if t % 16 == 0:
            sc = whole.station_scalars()
            sl = whole.slots()
            for k in (0, 1):
                cars = sl[k][:, 0, :].sum(axis=1)
                assert np.array_equal(cars, sc[:, k, 3])                     # car_number == occupied slots
                assert np.all(sc[:, k, 4] >= 0) and np.all(sc[:, k, 4] <= 10)  # line <= max_line (CHS.hpp:197)
                assert np.all(sc[:, k, 0] <= sc[:, k, 2] + 1e-3)             # min_power <= max_power
                assert np.all(sc[:, k, 1] <= sc[:, k, 2] + 1e-3)             # charge_power <= max_power
                occ = sl[k][:, 0, :] > 0
                assert np.all(sl[k][:, 4, :][occ] >= 25.0 - 1e-3) and np.all(sl[k][:, 4, :][occ] <= 100.0)
                assert np.all(sl[k][:, 4, :][~occ] == 0)
            assert np.all(o[:, -3] >= 0.1 - 1e-6) and np.all(o[:, -3] <= 1.0 + 1e-6)  # tank SOC bounds (HYD:111-112)
    assert whole.fcev_stuck_count() == 0
    # random policy, this hub: mean episode return is O(10); guards against silently dead dynamics
    assert 0 < ret.mean() < 100 and ret.std() > 0.1
    whole.close(); a.close(); b.close()


def test_env_test_known_answer():
    """The reference's own smoke test (test/env_test.py:14-49): hub [20 fast, 25 slow], seed_rand=False,
    random.seed(0); np.random.seed(0), action=None; known answer (env_c1_envtest.npz):
    episode return 34.858789741560585, final H2 SOC 0.1525.  The reference's constructor consumes draws of the
    default-seeded streams (station constructors, the 101-step H2 sweep, its own reset); the oracle (checker)
    replays that constructor order on the CPU and the resulting stream state is installed on the GPU."""
    chub = hub()
    g = orclib.load_golden("env_c1_envtest")
    kw = kwargs_of(g)
    env = orclib.OrcEnv(orclib.golden_config(g), ctor_seeds=(1, 1))
    env.reset(g["ctor_days"], g["ctor_z"])
    buf = np.zeros(140, dtype=np.uint8)
    orc.orc_rng_export_glibc128(orc.orc_env_rng(env.e), ptr(buf))
    ring = buf[4:128].view(np.uint32)
    front = (int(buf[0:4].view(np.int32)[0]) // 5 + 3) % 31
    state = np.concatenate([ring, [front], buf[128:132].view(np.uint32)]).astype(np.uint32)[None, :]
    v = chub.VecChargingHub(1, rng="compat", **kw)
    v.set_telemetry(True)
    close(v.hy_table(), g["hy_table"], "hy_table", rtol=1e-13, atol=1e-12)
    v.set_compat_seeds([[1, 1]])
    v.reset(g["ctor_days"][None], g["ctor_z"][None])      # OU states as the constructor's reset leaves them
    v.set_compat_state(state)
    v.reset(g["reset_days"][0][None], g["reset_z"][0][None])
    close(v.obs_f64()[0], g["reset_obs"][0], "reset obs")
    ret = 0.0
    for i in range(96):
        obs, rew, done, _ = v.step(g["action"][i][None], g["exo_z"][i][None])
        sl = v.slots()
        check_slots(sl[0][0], g["slots0"][i], ("c1", i, 0))
        check_slots(sl[1][0], g["slots1"][i], ("c1", i, 1))
        close(v.obs_f64()[0], g["obs"][i], ("c1 obs", i), rtol=TIGHT, atol=TIGHT)
        close(v.reward_f64()[0], g["reward"][i], ("c1 reward", i), rtol=TIGHT, atol=TIGHT)
        ret += float(v.reward_f64()[0])
    assert abs(ret - 34.858789741560585) < 1e-9
    assert abs(v.telemetry()[0, 3] - 0.1525) < 1e-9
    v.close()


def _check_dropin_attributes(env, g, i, what):
    """every attribute the reference class exposes after step() (recorded per step by oracle/gen/gen_env_golden.py: `attrs`,
    `real_state`, `action_real`) on the drop-in class"""
    names = [str(x) for x in g["attr_names"]]
    want = dict(zip(names, g["attrs"][i]))
    for name in names:
        if name[-2:] in ("_0", "_1") and name[:-2] in ("re_income_evs_list", "re_income_evs_cost_list"):
            got = getattr(env, name[:-2])[int(name[-1])]
        elif name == "test_penalty" and np.isnan(want[name]):
            assert not hasattr(env, "test_penalty"), (what, i, "test_penalty exists only from the first episode end on (MGR:290)")
            continue
        else:
            got = getattr(env, name)
        close(float(got), want[name], (what, name, i), rtol=TIGHT, atol=TIGHT)
    close(env.real_state, g["real_state"][i], (what, "real_state", i), rtol=TIGHT, atol=TIGHT)
    assert np.array_equal(np.asarray(env.action_real, dtype=np.float64), g["action_real"][i]), (what, "action_real", i)
    # the sub-objects scripts reach into, through the very expressions the fixture generator applied to the reference object
    # (oracle/gen/gen_env_golden.py: telemetry(), station_block()): env.hy_sys / .sty / .hvs, env.hfc, env.env_aggregator.evcssp_evs_objects
    h = env.hy_sys
    tel = [env.hy_act, h.hy_flow_speed, h.all_power_second, h.sty.Store_SOC, h.sty.capacity, h.hvs.total_mass_need, h.sty.hy_use,
           h.sty.not_meet, env.fc_power, env.hfc.hy_to_use, env.re_used_renew, env.re_ev_power_list[0], env.re_ev_power_list[1],
           env.re_hydrogen_power, env.income, g["reward"][i], env.re_pv_power, env.re_wd_power, float(env.real_state[1]),
           h.hvs.arrive_number, h.hvs.line, len(h.hvs.needed_time_list)]
    close(tel[:19], g["telem"][i][:19], (what, "sub-object telemetry", i), rtol=TIGHT, atol=1e-7)
    assert tel[19:] == list(g["telem"][i][19:22]), (what, "fcev ints", i, tel[19:], g["telem"][i][19:22])
    blk = []
    for st in env.env_aggregator.evcssp_evs_objects:
        blk += [st.min_power, st.charge_power, st.max_power, st.car_number, st.line, st.flow_in_number[-1]]
    assert np.array_equal(np.array(blk), g["stations"][i]), (what, "stations", i, blk, g["stations"][i])
    ag = env.env_aggregator
    assert ag.evcssp_charge_power == [blk[1], blk[7]] and ag.ag_flow_in_number == [blk[5], blk[11]] and ag.aggregator_time_hole == env.real_state[0]
    assert len(ag.price) == 96 + (i % int(g["steps_per_episode"])) + 1 and h.sys_time == env.real_state[0]  # AGG:147,171


def test_dropin_class_reproduces_env_test():
    """test/env_test.py through the drop-in class, nothing injected: EvcsspManagerEnv_v6(**env_kwargs) with
    seed_rand=False after random.seed(0); np.random.seed(0), reset(), step(action=None) until done.
    Reference: return 34.858789741560585, final H2 SOC 0.1525, the recorded per-step obs / rewards, and every attribute the
    reference class carries after a step (cumulated_draw_ele, real_state, test_penalty at the episode end, re_* ...)."""
    import random
    chub = hub()
    g = orclib.load_golden("env_c1_envtest")
    random.seed(0)
    np.random.seed(0)
    env = chub.EvcsspManagerEnv_v6(station_list=[20, 25], station_type_list=["fast", "slow"], constant_charging=False,
                                   seed_rand=False, hydro_prod_rate=100, hydro_store_vlt=500 / 20, init_soc=0.2,
                                   fc_max_power=100, fcev_permeate=0.01, use_lagrange=False, renew_fluctuate=0.0,
                                   price_fluctuate=0.0, hydro_loss=0.0)
    assert env.observation_space.shape == (13,) and env.action_space.shape == (47,)
    assert env.real_state == [] and env.state is None and env.penalty == 0 and env.cumulated_draw_ele == 0  # MGR:121-130
    o = env.reset()
    close(o, g["reset_obs"][0], "drop-in reset obs", rtol=TIGHT, atol=TIGHT)
    close(env.real_state, g["reset_real_state"][0], "drop-in reset real_state", rtol=TIGHT, atol=TIGHT)
    assert [env.cumulated_income, env.cumulated_draw_ele, env.penalty] == list(g["reset_attrs"][0]) and env.lagrangian_factor is None
    done, ret, i = False, 0.0, 0
    while not done:
        s_, r, done, info = env.step(action=None)
        close(s_, g["obs"][i], ("drop-in obs", i), rtol=TIGHT, atol=TIGHT)
        close(r, g["reward"][i], ("drop-in reward", i), rtol=TIGHT, atol=TIGHT)
        assert isinstance(r, float) and isinstance(done, bool) and info == {}
        _check_dropin_attributes(env, g, i, "c1")
        ret += r
        i += 1
    assert i == 96
    assert abs(ret - 34.858789741560585) < 1e-9
    assert abs(env._t_Store_SOC - 0.1525) < 1e-12
    assert env.test_penalty > 0 and env.cumulated_draw_ele > 0
    env.close()


# the seed the fixture's generator gave Python's and numpy's global generators in front of the constructor (oracle/gen/gen_env_golden.py:
# main): the drop-in class draws its exogenous variates from them where the reference does
PY_SEEDS = {"env_c3_random": 1, "env_c2_random": 2, "env_c5_random": 3, "env_slow_only_fcev": 4, "env_clamp": 5, "env_full_tank": 6,
            "env_constant": 7, "env_fcev_queue": 8, "env_small_fast_neg": 9, "env_fcev_queue_deep": 10, "env_big_100_70": 11,
            "env_slow_slow": 12, "env_fast_fast": 13, "env_no_electrolyser": 14, "env_permeate_cap": 15, "env_one_pile": 16,
            "env_constant_swapped": 17, "env_past_done": 18, "env_past_done_c2": 19, "env_defaults": 20, "env_tank_floor": 21, "env_tank_brim": 22}


@pytest.mark.parametrize("name", sorted(PY_SEEDS))
def test_dropin_class_reproduces_reference_fixture(name):
    """the drop-in class under the recorded random actions of EVERY fixture (hub shapes from one pile per station to 100 + 70, stations
    without piles, swapped and equal station kinds, the constant-power fleet, no electrolyser, stuck forecourts; episodes back to back,
    re-seeded or not, cut short or not, as recorded): observations, rewards and every reference attribute, step by step.  The
    reference's two C++ streams are process globals seeded by the fixture's generator in front of the constructor and of the
    reset()s; the drop-in class takes the same seeds for this env's own streams (compat_seeds, set_compat_seeds)."""
    import random
    chub = hub()
    g = orclib.load_golden(name)
    random.seed(PY_SEEDS[name])
    np.random.seed(PY_SEEDS[name])
    kw = kwargs_of(g)
    if "ctor_kwargs_names" in g.files:  # what the reference's constructor was actually given: everything else is left to the drop-in's defaults too
        given = set(str(x) for x in g["ctor_kwargs_names"])
        kw = {k: v for k, v in kw.items() if k in given}
    env = chub.EvcsspManagerEnv_v6(seed_rand=False, use_lagrange=False, compat_seeds=[int(x) for x in g["ctor_seeds"]], **kw)
    steps, i = int(g["steps_per_episode"]), 0
    seeds = {int(r[0]): (int(r[1]), int(r[2])) for r in g["seeds"]}
    for ep in range(int(g["episodes"])):
        if ep in seeds:
            env.set_compat_seeds(*seeds[ep])
        o = env.reset()
        close(o, g["reset_obs"][ep], (name, "reset obs", ep), rtol=TIGHT, atol=TIGHT)
        close(env.real_state, g["reset_real_state"][ep], (name, "reset real_state", ep), rtol=TIGHT, atol=TIGHT)
        assert [env.cumulated_income, env.cumulated_draw_ele, env.penalty] == list(g["reset_attrs"][ep])
        for t in range(steps):
            s_, r, done, info = env.step(g["action"][i])
            close(s_, g["obs"][i], (name, "obs", i), rtol=TIGHT, atol=TIGHT)
            close(r, g["reward"][i], (name, "reward", i), rtol=TIGHT, atol=TIGHT)
            assert done == bool(g["done"][i])
            _check_dropin_attributes(env, g, i, name)
            i += 1
    env.close()


@pytest.mark.parametrize("rng", ["philox", "compat"])
def test_snapshot_restore_roundtrip(rng):
    """chub_get_state / chub_set_state: branching from a snapshot replays the same future bit for bit."""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.05)
    n = 96
    v = chub.VecChargingHub(n, seed=5, rng=rng, **kw)
    rs = np.random.RandomState(0)
    days = np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1) if rng == "compat" else None
    z = (lambda: rs.normal(size=(n, 3))) if rng == "compat" else (lambda: None)
    v.reset(days, z())
    acts = [rs.uniform(-1, 1, size=(n, v.act_dim)).astype(np.float32) for _ in range(40)]
    zs = [z() for _ in range(40)]
    for t in range(10):
        v.step(acts[t], zs[t])
    snap = v.get_state()
    first = [v.step(acts[t], zs[t]) for t in range(10, 40)]
    slots_a = v.slots()
    v.set_state(snap)
    assert v.clock == 10
    second = [v.step(acts[t], zs[t]) for t in range(10, 40)]
    slots_b = v.slots()
    for a, b in zip(first, second):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(slots_a, slots_b))
    other = chub.VecChargingHub(n + 1, seed=5, rng=rng, **kw)
    with pytest.raises(chub.ChubError):
        other.set_state(snap)
    other.close()
    v.close()


@pytest.mark.parametrize("cc", [False, True])
@pytest.mark.parametrize("rng", ["philox", "compat"])
@pytest.mark.parametrize("piles", [(20, 25), (100, 70), (3, 170)], ids=["20_25", "100_70", "3_170"])
def test_scalar_load_mode_matches_oracle(piles, rng, cc):
    """evs_step(float) on the GPU against the oracle, whose station code for this mode is pinned
    bit for bit against the reference (tests/test_oracle_vs_ref.py::test_station_scalar_load_mode, up to 170 piles).  Stations of
    more than 64 piles: a unit is a workgroup of its own (k_slot_unit) and the multimap order of assign_on_off
    (CHS.hpp:1318-1362 / 1629-1674) is a rank over the whole workgroup."""
    chub = hub()
    kw = dict(station_list=list(piles), station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, constant_charging=cc, renew_fluctuate=0.0,
              price_fluctuate=0.0, hydro_loss=0.0)
    n, seed = (70, 99) if piles == (20, 25) else (9, 99)
    # (COMPAT handles have two launch forms, chosen by batch size: the constant-power run takes the split step of large batches, the other one
    # kernel per station)
    form = ("packed" if cc else "wave") if rng == "compat" and piles == (20, 25) else "auto"
    v = chub.VecChargingHub(n, seed=seed, rng=rng, slot_kernel=form, **kw)
    v.set_telemetry(True)
    cfg = orclib.make_config(piles=kw["station_list"], types=kw["station_type_list"], constant_charging=cc,
                             hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    mode = orclib.PHILOX if rng == "philox" else orclib.COMPAT
    h = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 0, mode, seed)
    rs = np.random.RandomState(4)
    o_obs, o_rew, o_done = np.zeros((n, v.obs_dim)), np.zeros(n), np.zeros(n, dtype=np.uint8)
    if rng == "compat":
        days = np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32)
        z = rs.normal(size=(n, 3))
        v.reset(days, z)
        orc.orc_vec_reset(h, ptr(days), ptr(z), ptr(o_obs))
    else:
        v.reset()
        orc.orc_vec_reset(h, None, None, ptr(o_obs))
    for t in range(120):
        sc = v.station_scalars()
        loads = np.stack([rs.uniform(0, 1.2, n) * (sc[:, 0, 2] + 1.0), rs.uniform(0, 1.2, n) * (sc[:, 1, 2] + 1.0)], axis=1)
        loads = loads.astype(np.float32)
        tail = rs.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        z = rs.normal(size=(n, 3)) if rng == "compat" else None
        act = v.load_actions(loads, tail)
        obs, rew, done, _ = v.step_load(loads, tail, z)
        orc.orc_vec_step_load(h, ptr(act), ptr(z) if z is not None else None, ptr(o_obs), ptr(o_rew), ptr(o_done), 2)
        sl = v.slots()
        sc = v.station_scalars()
        for e in range(n):
            env = orc.orc_vec_env(h, e)
            for k, nk in ((0, piles[0]), (1, piles[1])):
                want = np.zeros((9, nk), dtype=np.float32)
                orc.orc_station_slots(orc.orc_env_station(env, k), ptr(want))
                check_slots(sl[k][e], want, ("load mode", rng, cc, t, e, k))
                ws = np.zeros(8)
                orc.orc_station_scalars(orc.orc_env_station(env, k), ptr(ws))
                assert np.array_equal(sc[e, k, :6], ws[:6]), (t, e, k, sc[e, k], ws)
        close(v.obs_f64(), o_obs, ("load mode obs", t), rtol=TIGHT, atol=TIGHT)
        close(v.reward_f64(), o_rew, ("load mode reward", t), rtol=TIGHT, atol=TIGHT)
        if t == 95:
            if rng == "compat":
                v.reset(days, np.zeros((n, 3)))
                orc.orc_vec_reset(h, ptr(days), ptr(np.zeros((n, 3))), ptr(o_obs))
            else:
                v.reset()
                orc.orc_vec_reset(h, None, None, ptr(o_obs))
    orc.orc_vec_destroy(h)
    v.close()


def test_vector_adapters_and_info_telemetry():
    """the stable-baselines / Gymnasium conventions over the HIP hub give the same trajectory as the
    plain VecChargingHub loop, with the reference's re_* accounting in info"""
    chub = hub()
    from charginghub_env_amd import wrappers
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    n, seed = 64, 99
    plain = chub.VecChargingHub(n, seed=seed, **kw)
    plain.set_telemetry(True)
    sb3 = wrappers.HubVecEnv(n_envs=n, seed=seed, telemetry=True, **kw)
    gym = wrappers.HubVectorEnv(n_envs=n, seed=seed, telemetry=True, **kw)
    assert sb3.observation_space.shape == (13,) and sb3.action_space.shape == (47,)
    o0 = plain.reset()
    assert np.array_equal(o0, sb3.reset()) and np.array_equal(o0, gym.reset()[0])
    rs = np.random.RandomState(5)
    for t in range(100):
        a = rs.uniform(-1, 1, (n, 47)).astype(np.float32)
        if t == 96:
            g = gym.step(a)                      # Gymnasium "next step" autoreset consumes one call
            assert np.array_equal(g[0], o_reset) and not g[2].any()
        o, r, d, _ = plain.step(a)
        so, sr, sd, si = sb3.step(a)
        go, gr, gt, gtr, gi = gym.step(a)
        assert np.array_equal(r, sr) and np.array_equal(r, gr) and np.array_equal(d, sd) and np.array_equal(d, gt)
        assert np.array_equal(o, go)
        tel = plain.telemetry()
        assert np.array_equal(gi["re_used_renew"], tel[:, 10]) and np.array_equal(gi["income"], tel[:, 14])
        assert si[3]["income"] == tel[3, 14] and np.array_equal(si[3]["re_ev_power_list"], tel[3, 11:13])
        if d.all():
            assert t == 95
            assert np.array_equal(si[0]["terminal_observation"], o[0]) and si[0]["TimeLimit.truncated"] is False
            o_reset = plain.reset()
            assert np.array_equal(so, o_reset)
        else:
            assert np.array_equal(o, so)
    for x in (plain, sb3, gym):
        x.close()


def test_make_time_limit_on_dropin():
    """gym.make('evcssp_env_cpp:charging-hub-v6', **env_kwargs) of test/env_test.py:36 without gym"""
    import random
    chub = hub()
    random.seed(0)
    np.random.seed(0)
    env = chub.make("evcssp_env_cpp:charging-hub-v6", station_list=[20, 25], station_type_list=["fast", "slow"],
                    constant_charging=False, seed_rand=False, hydro_prod_rate=100, hydro_store_vlt=25, init_soc=0.2,
                    fc_max_power=100, fcev_permeate=0.01, use_lagrange=False, renew_fluctuate=0, price_fluctuate=0,
                    hydro_loss=0)
    env.reset()
    done, ret, steps = False, 0.0, 0
    while not done:
        s_, r, done, info = env.step(action=None)
        ret += r
        steps += 1
    assert steps == 96 and info == {}
    assert abs(ret - 34.858789741560585) < 1e-9       # the reference's own smoke run
    env.close()


def test_staggered_groups_on_gpu():
    """non-lock-step episodes: 4 groups a quarter day apart; every group is bit-identical to a plain
    hub over the same global env range that got the same head start"""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    n, G, seed = 64, 4, 31
    st = chub.StaggeredHub(n, G, seed=seed, env_id0=500, **kw)
    obs = st.reset()
    assert st.clocks == [0, 24, 48, 72]
    g = 2
    ref = chub.VecChargingHub(n // G, seed=seed, env_id0=500 + g * (n // G), **kw)
    o = ref.reset()
    head = np.zeros((n // G, 47), dtype=np.float32)
    head[:, :45] = 1.0
    for _ in range(48):
        o = ref.step(head)[0]
    rows = slice(g * (n // G), (g + 1) * (n // G))
    assert np.array_equal(obs[rows], o)
    rs = np.random.RandomState(3)
    ends = []
    for t in range(1, 110):
        a = rs.uniform(-1, 1, (n, 47)).astype(np.float32)
        obs, rew, done, info = st.step(a)
        o, r, d, _ = ref.step(a[rows])
        assert np.array_equal(rew[rows], r) and np.array_equal(done[rows], d)
        if d.all():
            assert np.array_equal(info["terminal_observation"][rows], o) and g in info["reset_groups"]
            o = ref.reset()
        assert np.array_equal(obs[rows], o)
        ends.append(sorted(set(np.nonzero(done)[0] // (n // G))))
    assert [t + 1 for t, e in enumerate(ends) if e] == [24, 48, 72, 96]
    assert [e for e in ends if e] == [[3], [2], [1], [0]]
    st.close()
    ref.close()


def test_determinism_and_long_run():
    """same seed -> the same trajectory on two handles (atomics and dense-queue order must not leak into results);
    ten episodes stay finite and in range, the FCEV forecourt never gets stuck at the default permeate"""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.3, price_fluctuate=0.3)
    n = 4096
    a_, b_ = chub.VecChargingHub(n, seed=77, **kw), chub.VecChargingHub(n, seed=77, **kw)
    rs = np.random.RandomState(9)
    oa, ob = a_.reset(), b_.reset()
    assert np.array_equal(oa, ob)
    for t in range(960):
        act = rs.uniform(-1, 1, (n, 47)).astype(np.float32)
        oa, ra, da, _ = a_.step(act)
        if t % 96 < 40 or t % 96 == 95:      # the second handle shadows part of every episode and every episode end
            ob, rb, db, _ = b_.step(act)
            assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db), t
            if t % 96 == 39:
                snap = a_.get_state()        # ... and is re-synchronised from a snapshot where it stops shadowing
        elif t % 96 == 94:
            b_.set_state(a_.get_state())
        assert np.isfinite(oa).all() and np.isfinite(ra).all() and np.abs(oa).max() < 50
        if da.all():
            assert t % 96 == 95
            oa, ob = a_.reset(), b_.reset()
            assert np.array_equal(oa, ob)
    assert a_.fcev_stuck_count() == 0
    sl = a_.slots()
    assert sl[0][:, 4].min() >= 0 and sl[0][:, 4].max() <= 100.0 + 1e-3 and sl[1][:, 4].max() <= 100.0 + 1e-3  # SoC in range
    a_.close()
    b_.close()


def test_error_paths_on_gpu():
    """error behaviour of the ABI with a device present: codes + messages, nothing crashes, the handle stays usable"""
    chub = hub()
    from charginghub_env_amd import ChubError
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    v = chub.VecChargingHub(8, seed=1, **kw)
    a = np.zeros((8, 47), dtype=np.float32)
    with pytest.raises(ChubError, match="before reset"):
        v.step(a)                                       # MGR would fail on its unset state too
    v.reset()
    with pytest.raises(AssertionError):
        v.step(a[:, :46])                               # MGR:148
    blob = v.get_state()
    with pytest.raises(ChubError):
        v.set_state(blob[:-8])                          # truncated snapshot
    other = chub.VecChargingHub(16, seed=1, **kw)
    with pytest.raises(ChubError):
        other.set_state(blob)                           # snapshot of a different batch size
    other.close()
    o1 = v.step(a)[0]
    v.set_state(blob)
    assert np.array_equal(v.step(a)[0], o1)             # still consistent after the failed calls
    c = chub.VecChargingHub(4, seed=1, rng="compat", **kw)
    with pytest.raises(ChubError, match="COMPAT"):
        c.reset()                                       # the reference's host draws must be supplied in this mode
    c.close()
    with pytest.raises(ChubError):
        chub.VecChargingHub(8, seed=1, device=99, **kw)
    with pytest.raises(ChubError, match="fused_step = 2"):
        chub.VecChargingHub(8, seed=1, tile="large", fused_step="on", **kw)  # the single-launch step runs on the small tile
    big = chub.VecChargingHub(8, seed=1, tile="large", **kw)
    assert big.uses_packed_kernel and not big.uses_fused_step              # "auto" would have fused a handle this small
    big.close()
    with pytest.raises(ChubError, match="too many tape classes"):
        v.tape_register_soc(np.linspace(25.0, 70.0, 2049, dtype=np.float32))  # the state word has 11 bits for the class
    with pytest.raises(ChubError, match="out of range"):
        rows = np.full((8, 45, 6), -1, dtype=np.int32)
        rows[0, 0] = [0, 0, 40, 0, 0, 0]                                   # a stay of 40 slots does not fit the 5-bit fields
        v.set_slots(rows)
    v.close()


STAT_HUBS = {
    "c2": (dict(station_list=[16, 0], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                fc_max_power=100.0, fcev_permeate=0.0), 8192),
    "c3": (dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                fc_max_power=100.0, fcev_permeate=0.01), 8192),
    "c5": (dict(station_list=[32, 32], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.3, price_fluctuate=0.3), 4096),
}


@pytest.mark.parametrize("policy", ["random", "all_on", "all_off"])
@pytest.mark.parametrize("hub_name", sorted(STAT_HUBS))
def test_philox_is_statistically_the_reference_process_on_gpu(hub_name, policy):
    """PHILOX (production streams) against COMPAT (the reference's own streams, which reproduce the reference bit for bit) on the
    device: thousands of envs x one episode each under the same policy, on the C2, C3 and C5 hubs, under a random policy, every
    pile switched on, every pile switched off (only forced charging).  Every statistic is a per-env quantity, so the difference
    of the two modes' means is judged against ITS OWN standard error: |difference| <= K standard errors, no fixed tolerances.
    (The laws of the individual variates are bounded exactly, without sampling, in tests/test_law_fidelity_cpu.py; this is the
    whole process end to end.)"""
    chub = hub()
    K = 4.5  # 36 comparisons in all: P(any |z| > 4.5 under equality) < 3e-4
    kw, n = STAT_HUBS[hub_name]
    S = sum(kw["station_list"])
    rs = np.random.RandomState(5)
    acts = []
    for _ in range(96):
        a = rs.uniform(-1, 1, (n, S + 2)).astype(np.float32)
        if policy == "all_on":
            a[:, :S] = 1.0
        elif policy == "all_off":
            a[:, :S] = -1.0
        acts.append(a)
    out = {}
    for mode in ("compat", "philox"):
        v = chub.VecChargingHub(n, seed=4242, rng=mode, **kw)
        rz = np.random.RandomState(17)
        if mode == "compat":
            days = np.stack([rz.randint(0, 100, n), rz.randint(0, 150, n)], axis=1).astype(np.int32)
            v.reset(days, rz.normal(size=(n, 3)))
        else:
            v.reset()
        ret, cars, line, flow, power = (np.zeros(n) for _ in range(5))
        for t in range(96):
            o, r, d, _ = v.step(acts[t], rz.normal(size=(n, 3)) if mode == "compat" else None)
            ret += r
            sc = v.station_scalars()
            cars += sc[:, :, 3].sum(axis=1) / 96.0     # mean occupancy over the day
            line += sc[:, :, 4].sum(axis=1) / 96.0     # mean queue length
            flow += sc[:, :, 5].sum(axis=1)            # cars that arrived over the day
            power += sc[:, :, 1].sum(axis=1) / 96.0    # mean charging power
        out[mode] = dict(ret=ret, cars=cars, line=line, flow=flow, power=power, soc=o[:, -3].astype(np.float64))
        v.close()
    a, b = out["compat"], out["philox"]
    report = []
    worst = 0.0
    for key in ("ret", "cars", "line", "flow", "power", "soc"):
        diff = b[key].mean() - a[key].mean()
        se = np.sqrt(a[key].var(ddof=1) / n + b[key].var(ddof=1) / n)
        z = diff / se if se > 0 else 0.0
        worst = max(worst, abs(z))
        report.append("%s %.4f vs %.4f (diff %+.4f, se %.4f, z %+.2f)" % (key, a[key].mean(), b[key].mean(), diff, se, z))
    print("%s / %s, %d envs: " % (hub_name, policy, n) + "; ".join(report))
    for key in ("ret", "cars", "line", "flow", "power", "soc"):
        diff = b[key].mean() - a[key].mean()
        se = np.sqrt(a[key].var(ddof=1) / n + b[key].var(ddof=1) / n)
        assert abs(diff) <= K * se + 1e-12, (hub_name, policy, key, a[key].mean(), b[key].mean(), se)
    # spread of the episode return: the ratio of two sample variances of n values each has standard error ~ sqrt(2 * (kurt - 1) / n)
    ra, rb = a["ret"], b["ret"]
    kurt = 0.5 * (((ra - ra.mean()) ** 4).mean() / ra.var() ** 2 + ((rb - rb.mean()) ** 4).mean() / rb.var() ** 2)
    assert abs(np.log(rb.var() / ra.var())) <= K * np.sqrt(2.0 * max(kurt - 1.0, 1.0) / n), (ra.std(), rb.std(), kurt)


def test_graph_replay_equals_eager():
    """a hipGraph of two whole episodes (chub_graph_*), replayed twice, against the same four episodes issued call by call:
    the replays bake clocks and buffers in but not the random streams (the Philox tick base moves on with every replay), so
    the two runs are the same simulation bit for bit"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    n = 3000
    out = []
    for mode in ("eager", "graph"):
        v = chub.VecChargingHub(n, seed=2024, **kw)
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 5, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
        obs0 = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)

        def episode_pair():
            for i in range(192):
                if i % 96 == 0:
                    v.reset_device(obs0.ptr, stream=st.ptr)
                v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)

        if mode == "eager":
            episode_pair()
            episode_pair()
        else:
            st.sync()
            v.graph_begin(st.ptr)
            episode_pair()
            g = v.graph_end(st.ptr)
            v.graph_launch(g, st.ptr)
            v.graph_launch(g, st.ptr)
            st.sync()
            v.graph_destroy(g)
        last = packed[1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr)
        out.append((last, v.slots(), v.station_scalars()))
        # and the handle goes on eagerly from there, the same in both runs
        o = v.reset()
        out[-1] += (o, v.step(np.zeros((n, v.act_dim), dtype=np.float32))[0])
        v.close()
        st.destroy()
    (la, sa_, ca, oa, na), (lb, sb, cb, ob, nb) = out
    assert np.array_equal(la, lb) and np.array_equal(ca, cb) and all(np.array_equal(x, y) for x, y in zip(sa_, sb))
    assert np.array_equal(oa, ob) and np.array_equal(na, nb)
    assert (la[:, -1] == 1.0).all()                                    # the 96th step of an episode: done
    with pytest.raises(chub.ChubError):                                # an odd number of calls cannot be replayed
        v = chub.VecChargingHub(64, seed=1, **kw)
        st = multi_gpu.Stream(0)
        try:
            v.graph_begin(st.ptr)
            v.reset_device(obs0.ptr, stream=st.ptr) if False else v.reset_device(multi_gpu.DeviceBuffer(64 * v.obs_dim * 4).ptr, stream=st.ptr)
            v.graph_end(st.ptr)
        finally:
            v.close()


def test_graph_replays_mixed_with_eager_calls():
    """replay -> eager calls -> replay (the second replay used to run on ticks the eager calls had consumed), a
    graph captured in the middle of an episode (bench.py's form for short timed regions: W warm-up steps, then ONE graph of the
    K timed steps), and two graphs of one handle replayed alternately: every sequence equals the same calls issued one by one"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    n = 1500
    # a program is a list of segments ("g", name, first, count) = graph `name` covering steps first .. first+count-1 (captured at
    # its first use, replayed at every use), or ("e", first, count) = those steps as calls; step i resets first when i % 96 == 0
    programs = {
        "replay_eager_replay": [("g", "a", 0, 192), ("e", 0, 96), ("g", "a", 0, 192), ("e", 0, 96), ("e", 0, 96), ("g", "a", 0, 192), ("e", 0, 3)],
        "span_mid_episode": [("e", 0, 5), ("g", "s", 5, 20), ("e", 25, 71), ("e", 0, 1)],
        "span_with_reset_inside": [("e", 0, 90), ("g", "s", 90, 13), ("e", 103, 4)],
        "two_graphs": [("g", "a", 0, 190), ("e", 94, 2), ("g", "b", 0, 190), ("e", 94, 2), ("g", "a", 0, 190), ("e", 94, 2), ("g", "b", 0, 190)],
    }
    for name, prog in programs.items():
        res = []
        for mode in ("eager", "graph"):
            v = chub.VecChargingHub(n, seed=77, **kw)
            st = multi_gpu.Stream(0)
            acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
            for b, a in enumerate(acts):
                v.random_actions_device(a.ptr, 9, b, st.ptr)
            packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
            obs0 = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)

            def steps(first, count):
                for i in range(first, first + count):
                    if i % 96 == 0:
                        v.reset_device(obs0.ptr, stream=st.ptr)
                    v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)

            graphs = {}
            trace = []
            for seg in prog:
                if seg[0] == "e" or mode == "eager":
                    first, count = seg[-2], seg[-1]
                    steps(first, count)
                else:
                    _, gname, first, count = seg
                    if gname not in graphs:
                        st.sync()
                        v.graph_begin(st.ptr)
                        steps(first, count)
                        graphs[gname] = v.graph_end(st.ptr)
                    v.graph_launch(graphs[gname], st.ptr)
                last_i = seg[-2] + seg[-1] - 1
                trace.append(packed[last_i & 1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
            trace.append(np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1))
            trace.append(v.station_scalars().reshape(n, -1))
            res.append(trace)
            if mode == "graph" and name == "replay_eager_replay":
                # the handle is 3 steps into a day: a replay of the graph captured at the start of a day would bake the wrong
                # clocks in -- refused, and nothing moves
                with pytest.raises(chub.ChubError, match="clock"):
                    v.graph_launch(graphs["a"], st.ptr)
                assert v.clock == 3
            for g in graphs.values():
                v.graph_destroy(g)
            v.close()
            st.destroy()
        for k, (a, b) in enumerate(zip(*res)):
            assert np.array_equal(a, b), (name, "segment", k)
        # no two episodes of a run are the same episode (a replay that reused ticks would repeat one)
        ends = [t for t in res[1][:-2]]
        for x in range(len(ends)):
            for y in range(x + 1, len(ends)):
                assert not np.array_equal(ends[x][:, :-2], ends[y][:, :-2]), (name, "segments", x, y, "ended in the same state")


def test_run_steps_issues_the_same_calls_from_c():
    """chub_run_steps (a span of steps issued from C: resets at the day boundaries, packed outputs double-buffered) against the
    same calls made one by one from Python: bit-identical outputs, state and clock -- with and without a hipGraph around it"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02)
    n = 1200
    res = []
    for form in ("python", "c", "c_in_graph"):
        v = chub.VecChargingHub(n, seed=12, **kw)
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(3)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 5, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (v.obs_dim + 2) * 4) for _ in range(2)]
        obs0 = multi_gpu.DeviceBuffer(n * v.obs_dim * 4)
        c_acts = (C.c_void_p * 3)(*[a.ptr for a in acts])
        c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)

        def span(first, count):
            if form == "python":
                for i in range(first, first + count):
                    if i % 96 == 0:
                        v.reset_device(obs0.ptr, stream=st.ptr)
                    v.step_device_packed(acts[i % 3].ptr, packed[i & 1].ptr, stream=st.ptr)
            else:
                check(v._lib.chub_run_steps(v._h, None, c_acts, 3, c_packed, None, obs0.ptr, first, count, st.ptr))

        trace = []
        span(0, 100)
        trace.append(packed[1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        if form == "c_in_graph":
            st.sync()
            v.graph_begin(st.ptr)
            span(100, 93)          # ... up to the end of the second day: 93 steps, no reset inside: + 1 below = an even count
            span(193, 95 + 1 + 1)  # step 193 .. 289 crosses two day boundaries (resets at 288): 97 steps + 1 reset
            g = v.graph_end(st.ptr)
            v.graph_launch(g, st.ptr)
            st.sync()
            v.graph_destroy(g)
        else:
            span(100, 93)
            span(193, 97)
        trace.append(packed[(193 + 97 - 1) & 1].to_host(np.float32, (n, v.obs_dim + 2), st.ptr))
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), np.array([v.clock])]
        res.append(trace)
        v.close()
        st.destroy()
    for k in range(len(res[0])):
        assert np.array_equal(res[0][k], res[1][k]), ("python vs c", k)
        assert np.array_equal(res[0][k], res[2][k]), ("python vs c in a graph", k)
    assert res[0][-1][0] == (193 + 97) % 96


def test_full_size_c5_properties():
    """BASELINE.json configs[4] at its own size: 262 144 envs x hub [32 fast, 32 slow] with fluctuating price / PV / wind, one
    episode on the production (packed) kernel: 4-way shard independence (what the 8-GPU job relies on), the invariants the
    reference asserts (MGR:191,205; HYD:111-112), occupancy bookkeeping.  This is the size where 32-bit slot offsets and the
    arena are largest."""
    chub = hub()
    kw = dict(station_list=[32, 32], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01, renew_fluctuate=0.3, price_fluctuate=0.3)
    N, Q = 262144, 4
    whole = chub.VecChargingHub(N, seed=12345, rng="philox", **kw)
    assert whole.uses_packed_kernel
    parts = [chub.VecChargingHub(N // Q, seed=12345, rng="philox", env_id0=q * (N // Q), **kw) for q in range(Q)]
    rs = np.random.RandomState(3)
    acts = [rs.uniform(-1, 1, size=(N, whole.act_dim)).astype(np.float32) for _ in range(3)]
    o = whole.reset()
    assert np.array_equal(o, np.concatenate([p.reset() for p in parts]))
    ret = np.zeros(N)
    for t in range(96):
        act = acts[t % 3]
        o, r, d, _ = whole.step(act)
        po, pr = zip(*[p.step(act[q * (N // Q):(q + 1) * (N // Q)])[:2] for q, p in enumerate(parts)])
        assert np.array_equal(o, np.concatenate(po)) and np.array_equal(r, np.concatenate(pr)), t
        assert np.all(np.isfinite(o)) and np.all(np.isfinite(r))
        assert d.all() == (t == 95) and d.any() == (t == 95)
        ret += r
        if t in (0, 50, 95):
            sc = whole.station_scalars()
            sl = whole.slots()
            for k in (0, 1):
                assert np.array_equal(sl[k][:, 0, :].sum(axis=1), sc[:, k, 3])   # car_number == occupied slots
                assert np.all(sc[:, k, 4] >= 0) and np.all(sc[:, k, 4] <= 10)    # line <= max_line (CHS.hpp:197)
                assert np.all(sc[:, k, 0] <= sc[:, k, 2] + 1e-3) and np.all(sc[:, k, 1] <= sc[:, k, 2] + 1e-3)
                occ = sl[k][:, 0, :] > 0
                assert np.all(sl[k][:, 4, :][occ] >= 25.0 - 1e-3) and np.all(sl[k][:, 4, :][occ] <= 100.0)
            assert np.all(o[:, -3] >= 0.1 - 1e-6) and np.all(o[:, -3] <= 1.0 + 1e-6)   # tank SOC bounds (HYD:111-112)
            del sl
    assert whole.fcev_stuck_count() == 0
    assert 0 < ret.mean() < 200 and ret.std() > 0.1
    whole.close()
    for p in parts:
        p.close()


@pytest.mark.parametrize("fused", ["auto", "off", "on"])
def test_step_bits_on_the_device_in_every_launch_form(fused):
    """chub_step_bits_device on the packed slot kernel reads the decision bits themselves (no action rows): the two-launch step,
    the single-launch step, on per-env clocks after calls on subsets, and recorded into a graph -- always what chub_step_device
    does with the float rows the bits were packed from"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1)
    n = 1300
    a = chub.VecChargingHub(n, seed=5, fused_step=fused, **kw)
    b = chub.VecChargingHub(n, seed=5, fused_step=fused, **kw)
    assert a.uses_packed_kernel and a.uses_fused_step == (fused != "off")
    lib, st = b._lib, multi_gpu.Stream(0)
    A, D, W = a.act_dim, a.obs_dim, a.bit_words
    rs = np.random.RandomState(8)
    d_act = multi_gpu.DeviceBuffer(n * A * 4)
    d_bits, d_tail = multi_gpu.DeviceBuffer(n * W * 8), multi_gpu.DeviceBuffer(n * 2 * 4)
    outs = [[multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)] for _ in range(2)]

    def upload(act):
        bits, tail = b.pack_actions(act)
        check(lib.chub_copy_to_device(0, d_act.ptr, act.ctypes.data, act.nbytes, st.ptr))
        check(lib.chub_copy_to_device(0, d_bits.ptr, bits.ctypes.data, bits.nbytes, st.ptr))
        check(lib.chub_copy_to_device(0, d_tail.ptr, tail.ctypes.data, tail.nbytes, st.ptr))
        st.sync()

    def both_step():
        check(lib.chub_step_device(a._h, d_act.ptr, None, outs[0][0].ptr, outs[0][1].ptr, outs[0][2].ptr, st.ptr))
        check(lib.chub_step_bits_device(b._h, d_bits.ptr, d_tail.ptr, None, outs[1][0].ptr, outs[1][1].ptr, outs[1][2].ptr, st.ptr))

    def same(where):
        for x, y, dt, sh in zip(outs[0], outs[1], (np.float32, np.float32, np.uint8), ((n, D), (n,), (n,))):
            assert np.array_equal(x.to_host(dt, sh, st.ptr), y.to_host(dt, sh, st.ptr)), where
        assert all(np.array_equal(x, y) for x, y in zip(a.slots(), b.slots())), where
        assert np.array_equal(a.station_scalars(), b.station_scalars()), where

    a.reset_device(outs[0][0].ptr, stream=st.ptr)
    b.reset_device(outs[1][0].ptr, stream=st.ptr)
    for t in range(12):
        upload(rs.uniform(-1, 1, size=(n, A)).astype(np.float32))
        both_step()
        same(("lock-step", t))
    # calls on subsets put both handles on per-env clocks (float rows for both: the packed form has no masked entry point) ...
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    act = rs.uniform(-1, 1, size=(n, A)).astype(np.float32)
    for v in (a, b):
        v.reset_envs(mask.astype(bool))
        v.step_envs((np.arange(n) % 5 != 0), act)
    # ... on which a call on everybody is again the packed form's business
    for t in range(6):
        upload(rs.uniform(-1, 1, size=(n, A)).astype(np.float32))
        both_step()
        same(("per-env clocks", t))
    assert a.clock_groups > 1 and np.array_equal(a.env_clocks(), b.env_clocks())
    # recorded: four steps of the packed form as one graph, replayed twice, against the float form call by call
    for v, o in ((a, outs[0]), (b, outs[1])):
        v.reset_device(o[0].ptr, stream=st.ptr)
    upload(rs.uniform(-1, 1, size=(n, A)).astype(np.float32))
    b.graph_begin(st.ptr)
    for _ in range(4):
        check(lib.chub_step_bits_device(b._h, d_bits.ptr, d_tail.ptr, None, outs[1][0].ptr, outs[1][1].ptr, outs[1][2].ptr, st.ptr))
    g = b.graph_end(st.ptr)
    for rep in range(2):
        if rep:  # a replay starts from the clock of the capture: a new day for both
            for v, o in ((a, outs[0]), (b, outs[1])):
                v.reset_device(o[0].ptr, stream=st.ptr)
        b.graph_launch(g, st.ptr)
        for _ in range(4):
            check(lib.chub_step_device(a._h, d_act.ptr, None, outs[0][0].ptr, outs[0][1].ptr, outs[0][2].ptr, st.ptr))
        same(("graph", rep))
    b.graph_destroy(g)
    a.close()
    b.close()


@pytest.mark.parametrize("label", ["c3", "big_100_70", "one_pile", "max64"])
def test_step_bits_equals_step_on_the_thresholded_actions(label):
    """chub_step_bits (one bit per pile + the two tail floats over PCIe) against chub_step on the full f32 action rows: the same
    handle arguments, the same actions -- observation, reward, done, slot state and station records bit for bit
    (action_to_real, evcssp_manager.py:384-393: a pile is on iff f32((a + 1) / 2) >= 0.5, i.e. a >= -2^-25)"""
    chub = hub()
    kw, n = next((c[1], c[2]) for c in PHILOX_CASES if c[0] == label)
    a = chub.VecChargingHub(n, seed=99, **kw)
    b = chub.VecChargingHub(n, seed=99, **kw)
    S = a.n_slots
    assert b.bit_words == (S + 63) // 64
    rs = np.random.RandomState(4)
    assert np.array_equal(a.reset(), b.reset())
    pinned = b.pinned_bits()
    for t in range(40):
        act = rs.uniform(-1, 1, size=(n, a.act_dim)).astype(np.float32)
        edge = rs.randint(0, 6, size=(n, S))  # values at and around the threshold
        act[:, :S] = np.where(edge == 0, np.float32(-2.0 ** -25), np.where(edge == 1, np.nextafter(np.float32(-2.0 ** -25), np.float32(-1)), act[:, :S]))
        if t % 9 == 0:
            act[:, :S] = 1.0 if t % 2 else -1.0
        o1, r1, d1, _ = a.step(act)
        bits, tail = b.pack_actions(act, out=pinned if t % 2 else None)
        # the packing itself, against the definition: bit j of env e
        for e in (0, n - 1):
            for j in (0, S - 1, S // 2):
                assert bool((int(bits[e, j >> 6]) >> (j & 63)) & 1) == bool(np.float32((act[e, j] + np.float32(1)) / np.float32(2)) >= np.float32(0.5))
        o2, r2, d2, _ = b.step_bits(bits, tail)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2), (label, t)
    assert all(np.array_equal(x, y) for x, y in zip(a.slots(), b.slots()))
    assert np.array_equal(a.station_scalars(), b.station_scalars())
    with pytest.raises(AssertionError):
        b.step_bits(np.zeros((n, b.bit_words + 1), dtype=np.uint64), np.zeros((n, 2), dtype=np.float32))
    a.close()
    b.close()


def test_copy_outputs_false_returns_the_pinned_arrays():
    """copy_outputs=False: step() hands out the handle's own pinned arrays (no copies on the Python side) with the same values"""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
    n = 300
    a_, b_ = chub.VecChargingHub(n, seed=4, **kw), chub.VecChargingHub(n, seed=4, copy_outputs=False, **kw)
    rs = np.random.RandomState(1)
    assert np.array_equal(a_.reset(), b_.reset())
    first = None
    for t in range(100):
        act = rs.uniform(-1, 1, size=(n, 47)).astype(np.float32)
        oa, ra, da, _ = a_.step(act)
        ob, rb, db, _ = b_.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db) and db.dtype == np.bool_
        first = ob if first is None else first
        assert ob is first                                 # the same array object every step
    a_.close()
    b_.close()


def test_compat_split_step_is_bit_identical_through_restores_and_masked_calls():
    """COMPAT handles have two launch forms for what k_compat_small does not take: one kernel per station with the unit's first lane walking
    the env's streams, and (large batches) the split step -- the walks one ENV per lane between a count of the units' empty slots and one
    slot pass over both stations, the pass leaving the next step's counts.  Same program of calls on both (forced by chub_options.slot_kernel):
    whole-batch resets and steps, a snapshot restored into a FRESH handle (its counts have to be made again), steps and a reset of a
    subset of the envs (per-env clocks), scalar-load steps -- observations, rewards, slot state, station records and the streams
    themselves bit for bit."""
    chub = hub()
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02)
    n = 300
    out = {}
    for form in ("wave", "packed", "packed_own_walks"):  # (packed: the walk two steps ahead, k_slot_walk2; packed_own_walks: chub_options.walk_ahead = 1)
        rs = np.random.RandomState(12)
        fkw = dict(slot_kernel="packed", walk_ahead="off") if form == "packed_own_walks" else dict(slot_kernel=form)
        v = chub.VecChargingHub(n, rng="compat", **fkw, **kw)
        v.set_compat_seeds(np.stack([rs.randint(1, 2**31 - 1, n), rs.randint(1, 2**31 - 1, n)], axis=1).astype(np.uint32))
        v.compat_replay_constructor()
        trace = []

        def note(obs, rew=None):
            trace.append(np.array(obs))
            if rew is not None:
                trace.append(np.array(rew))

        days = lambda: np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32)
        act = lambda: rs.uniform(-1, 1, size=(n, v.act_dim)).astype(np.float32)
        note(v.reset(days(), rs.normal(size=(n, 3))))
        for t in range(30):
            o, r, d, _ = v.step(act(), rs.normal(size=(n, 3)))
            note(o, r)
        snap = v.get_state()
        v.close()
        v = chub.VecChargingHub(n, rng="compat", **fkw, **kw)
        v.set_state(snap)
        for t in range(10):
            o, r, d, _ = v.step(act(), rs.normal(size=(n, 3)))
            note(o, r)
        # ... and a restore IN PLACE, into a handle that has stepped on: its walks ran ahead of steps that now never come, its
        # empty-slot counts (empt / empt2) and stream buffers are those of another moment -- the same continuation must come out again
        snap2 = v.get_state()
        st_rs = rs.get_state()
        cont = []
        for t in range(4):
            cont.append(v.step(act(), rs.normal(size=(n, 3)))[:2])
        v.set_state(snap2)
        rs.set_state(st_rs)
        for t in range(4):
            o, r = v.step(act(), rs.normal(size=(n, 3)))[:2]
            assert np.array_equal(o, cont[t][0], equal_nan=True) and np.array_equal(r, cont[t][1], equal_nan=True), (form, "in-place restore", t)
            note(o, r)
        mask = (rs.uniform(size=n) < 0.4).astype(np.uint8)
        for t in range(5):
            o, r, d = v.step_envs(mask, act(), rs.normal(size=(n, 3)))[:3]
            note(np.asarray(o)[mask != 0], np.asarray(r)[mask != 0])
        other = (rs.uniform(size=n) < 0.3).astype(np.uint8)
        note(np.asarray(v.reset_envs(other, days(), rs.normal(size=(n, 3))))[other != 0])
        for t in range(5):
            o, r, d = v.step_envs(1 - mask, act(), rs.normal(size=(n, 3)))[:3]
            note(np.asarray(o)[mask == 0], np.asarray(r)[mask == 0])
        sc = v.station_scalars()
        for t in range(4):
            loads = np.stack([rs.uniform(0, 1.2, n) * (sc[:, 0, 2] + 1.0), rs.uniform(0, 1.2, n) * (sc[:, 1, 2] + 1.0)], axis=1).astype(np.float32)
            o, r, d = v.step_load_envs(np.ones(n, dtype=np.uint8), loads, rs.uniform(-1, 1, size=(n, 2)).astype(np.float32), rs.normal(size=(n, 3)))[:3]
            note(o, r)
        note(v.reset(days(), rs.normal(size=(n, 3))))
        for t in range(6):
            o, r, d, _ = v.step(act(), rs.normal(size=(n, 3)))
            note(o, r)
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), v.compat_state()]
        out[form] = trace
        v.close()
    assert len(out["wave"]) == len(out["packed"]) == len(out["packed_own_walks"])
    for k, (a, b, c) in enumerate(zip(out["wave"], out["packed"], out["packed_own_walks"])):
        assert np.array_equal(a, b, equal_nan=True), k
        assert np.array_equal(a, c, equal_nan=True), ("own walks", k)


@pytest.mark.parametrize("label", ["c2", "c3_small", "ragged", "many_workgroups"])
def test_spans_of_steps_in_one_launch_are_bit_identical(label):
    """chub_run_steps on a handle that runs the one-launch step issues spans of lock-step steps as ONE launch (chub_options.span_steps), the per-env
    tails either on the workgroup's last slot wave behind its slot phases (k_steps_fused: span_tails = 1) or on a fifth wave ONE STEP BEHIND the
    slot waves (k_steps_piped: span_tails = 2 -- the tails of step s run beside the slot phases of step s + 1, the station records travel through
    two LDS buffers, the last step's tails run after the slot waves have ended).  The same program of calls with spans of any length, of at most
    7 steps (so that spans end and begin everywhere), in both forms, and with every step a launch of its own (span_steps = 1: the form every
    other test pins to the oracle / the reference): both packed blocks after every call, slot state, station records, clocks bit for bit --
    across day boundaries, from odd first steps, with 3 action batches (i % 3 against i & 1), mixed with single steps, a call on a subset of the
    envs (per-env clocks: no spans from there until everybody is reset) and inside a hipGraph.  "many_workgroups": 300 workgroups, where the
    default is the last slot wave and the tail wave is forced."""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    kw = {"c2": dict(station_list=[16, 0], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                     fc_max_power=100.0, fcev_permeate=0.0),
          "c3_small": dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                           fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.2, price_fluctuate=0.1),
          "ragged": dict(station_list=[7, 13], station_type_list=["slow", "fast"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.3,
                         fc_max_power=100.0, fcev_permeate=0.05, renew_fluctuate=0.3, price_fluctuate=0.3, hydro_loss=0.001),
          "many_workgroups": dict(station_list=[24, 9], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
                                  fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.1, price_fluctuate=0.2)}[label]
    n = {"c2": 4096, "c3_small": 1500, "ragged": 777, "many_workgroups": 4500}[label]
    res = {}
    FORMS = [("off", "auto"), ("auto", "auto"), ("auto", "same_wave"), ("auto", "own_wave"), (7, "same_wave"), (7, "own_wave"), ("graph", "same_wave"),
             ("graph", "own_wave")]
    for form in FORMS:
        v = chub.VecChargingHub(n, seed=21, span_steps="auto" if form[0] == "graph" else form[0], span_tails=form[1], **kw)
        assert v.uses_fused_step
        D, A = v.obs_dim, v.act_dim
        st = multi_gpu.Stream(0)
        acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(3)]
        for b, a in enumerate(acts):
            v.random_actions_device(a.ptr, 9, b, st.ptr)
        packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
        obs0, rew, done = multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)
        c_acts = (C.c_void_p * 3)(*[a.ptr for a in acts])
        c_packed = (C.c_void_p * 2)(packed[0].ptr, packed[1].ptr)
        trace = []

        def run(first, count):
            check(v._lib.chub_run_steps(v._h, None, c_acts, 3, c_packed, None, obs0.ptr, first, count, st.ptr))

        def note():
            trace.extend([packed[0].to_host(np.float32, (n, D + 2), st.ptr), packed[1].to_host(np.float32, (n, D + 2), st.ptr)])

        run(0, 5)       # reset + a short span
        note()
        run(5, 1)       # a single step through the same entry point
        note()
        v.step_device_packed(acts[0].ptr, packed[0].ptr, stream=st.ptr)  # step 6, from the host
        run(7, 96 - 7 + 30)  # an odd first step, to the day's end, a reset inside, 30 steps into the next day
        note()
        if form[0] == "graph":
            st.sync()
            v.graph_begin(st.ptr)
            run(126, 66 + 96)  # to the end of day 2 and through day 3: 162 steps + 1 reset = 163 launches' worth of ticks
            run(288, 2)        # + 1 reset + 2 steps: an even count of ticks in the graph (the draws are double-buffered by tick parity)
            g = v.graph_end(st.ptr)
            v.graph_launch(g, st.ptr)
            st.sync()
            v.graph_destroy(g)
        else:
            run(126, 66 + 96)
            run(288, 2)
        note()
        m = np.ascontiguousarray(np.arange(n) % 3 == 0, dtype=np.uint8)  # per-env clocks from here: run_steps goes step by step
        check(v._lib.chub_step_envs_device(v._h, m.ctypes.data, acts[1].ptr, None, obs0.ptr, rew.ptr, done.ptr, st.ptr))
        trace.append(obs0.to_host(np.float32, (n, D), st.ptr))
        run(290, 3)
        note()
        run(288 + 96, 40)  # (a multiple of 96: everybody is reset, one clock again, spans again)
        note()
        run(5, 80)  # first_step need not be the handle's clock (the caller may step on past `done`, MGR:271-299): the handle stands at slot 40, the
        note()      # span ends where ITS clock wraps (after 56 steps), the remaining 24 steps follow in the next day without a reset
        trace += [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), v.env_clocks(ticks=True)[0],
                  v.env_clocks(ticks=True)[1]]
        res[form] = trace
        v.close()
        st.destroy()
    for form in FORMS[1:]:
        assert len(res[form]) == len(res[FORMS[0]])
        for k, (a, b) in enumerate(zip(res[FORMS[0]], res[form])):
            assert np.array_equal(a, b), (label, "spans", form, "vs every step a launch: array", k)


def test_one_launch_forms_on_random_hub_shapes():
    """tools/experiments/shape_sweep.py on 14 random hub shapes and batch sizes (stations of 0 .. 64 piles of both kinds in both places, hubs of
    fewer than 8 piles, 37 .. 5000 envs): every step a launch with the one-launch step forced (k_step_tailwave / k_step_fused), spans with the
    tails on the last slot wave (k_steps_fused), on a wave of their own (k_steps_piped; refused below 8 piles) and of at most 5 steps -- against
    the two-launch step (k_slot_packed + k_env), every packed block, slot state and station records bit for bit."""
    import os
    import runpy
    import sys
    argv = sys.argv
    sys.argv = ["shape_sweep.py", "14", "3"]
    try:
        with pytest.raises(SystemExit) as ex:
            runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "experiments", "shape_sweep.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert ex.value.code == 0
