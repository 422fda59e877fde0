"""The launch forms against each other AT THE SIZES THEY ARE BENCHMARKED ON: what the small-handle form tests of
test_gpu_parity.py hold bit for bit -- the reference-exact COMPAT step in its three forms (one kernel per station with the unit's first
lane walking the env's streams; the split step with the stream walks two steps ahead, k_slot_walk2, walk workgroups [0, N / 64); the split
step walking for itself) and the PHILOX packed kernels' work orders and tiles -- must also hold where the grid is thousands of workgroups:
a grid-dependent index that is wrong only beyond a few hundred envs would pass every small test and still print a rate.

The reference behaviour at stake is the consumption order of the two process-global streams (CHS.hpp:1272-1316, 1583-1627;
hydro_sys.py:250-260), per env: the "wave" form IS that order on one lane (pinned against the reference fixtures at 1-6 envs,
test_compat_matches_reference_golden); the other forms must equal it at every env of a full-size handle with per-env distinct seeds.
Every step's observations / rewards / done flags are compared by digest (what differs first is named by its call number), the end state --
slot state, station records, the streams themselves -- array for array."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HUB_KW = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
              init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)


def hub():
    import charginghub_env_amd as chub
    return chub


def digest(*arrays):
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def compat_program(n, form, seed=2026, kw=HUB_KW):
    """One whole day, the next day's first step without a reset in between (the reference steps on past `done`, MGR:271-299), the episode-end
    reset, three steps, one call on a subset of the envs (per-env clocks from there on), two more calls on everybody.  -> (per-call digests,
    end state arrays, checksum of the end state)"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    fkw = dict(slot_kernel="packed", walk_ahead="off") if form == "packed_own_walks" else dict(slot_kernel=form)
    v = chub.VecChargingHub(n, rng="compat", **fkw, **kw)
    lib, h = v._lib, v._h
    rs = np.random.RandomState(seed)
    v.set_compat_seeds(np.stack([rs.randint(1, 2**31 - 1, n), rs.randint(1, 2**31 - 1, n)], axis=1).astype(np.uint32))
    v.compat_replay_constructor()
    D, A = v.obs_dim, v.act_dim
    st = multi_gpu.Stream(0)
    acts, zs = [], []
    for b in range(4):
        a = multi_gpu.DeviceBuffer(n * A * 4)
        v.random_actions_device(a.ptr, 4711, b, st.ptr)
        z = multi_gpu.DeviceBuffer(n * 3 * 8)
        z.from_host(rs.normal(size=(n, 3)), st.ptr)
        acts.append(a)
        zs.append(z)
    days = [multi_gpu.DeviceBuffer(n * 2 * 4) for _ in range(2)]
    for d in days:
        d.from_host(np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32), st.ptr)
    obs, rew, done = multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)
    calls = []

    def note(with_reward=True):
        o = obs.to_host(np.float32, (n, D), st.ptr)
        if with_reward:
            calls.append(digest(o, rew.to_host(np.float32, (n,), st.ptr), done.to_host(np.uint8, (n,), st.ptr)))
        else:
            calls.append(digest(o))

    def step(i):
        v.step_device(acts[i % 4].ptr, obs.ptr, rew.ptr, done.ptr, d_exo_z=zs[(i + 1) % 4].ptr, stream=st.ptr)
        note()

    v.reset_device(obs.ptr, days[0].ptr, zs[0].ptr, stream=st.ptr)
    note(False)
    for i in range(97):  # the day + the next day's first step, no reset in between
        step(i)
    assert done.to_host(np.uint8, (n,), st.ptr).sum() == 0  # (the 97th step is the first of a new day: not done)
    v.reset_device(obs.ptr, days[1].ptr, zs[2].ptr, stream=st.ptr)
    note(False)
    for i in range(3):
        step(i + 1)
    mask = np.ascontiguousarray(rs.uniform(size=n) < 0.4, dtype=np.uint8)
    check(lib.chub_step_envs_device(h, mask.ctypes.data, acts[2].ptr, zs[3].ptr, obs.ptr, rew.ptr, done.ptr, st.ptr))
    note()
    for i in range(2):
        step(i + 2)
    end = [np.concatenate([x.reshape(n, -1) for x in v.slots()], axis=1), v.station_scalars().reshape(n, -1), v.compat_state(), v.env_clocks()]
    v.close()
    st.destroy()
    return calls, end


@pytest.mark.parametrize("n", [4096, 65536])
def test_compat_launch_forms_are_bit_identical_at_bench_size(n):
    """65 536 envs x hub [20,25] is the size `roofline_compat` is quoted on (bench.py: 1024 walk workgroups in front of 5958 slot
    workgroups in k_slot_walk2's grid), 4096 envs the size of its second rate."""
    ref_calls, ref_end = compat_program(n, "wave")
    assert len(set(ref_calls)) == len(ref_calls)  # (no two calls alike: the digests see the state move)
    for form in ("packed", "packed_own_walks"):
        calls, end = compat_program(n, form)
        assert len(calls) == len(ref_calls)
        first = [k for k, (a, b) in enumerate(zip(ref_calls, calls)) if a != b]
        assert not first, (form, "first differing call", first[0], "of", len(calls))
        for k, (a, b) in enumerate(zip(ref_end, end)):
            assert np.array_equal(a, b, equal_nan=True), (form, "end state array", k)


@pytest.mark.parametrize("piles,types", [((24, 9), ("fast", "slow")), ((13, 8), ("slow", "fast")), ((64, 64), ("fast", "slow")), ((32, 32), ("slow", "slow")),
                                         ((8, 25), ("fast", "fast")), ((63, 10), ("slow", "fast")), ((25, 20), ("fast", "slow"))])
def test_compat_launch_forms_are_bit_identical_on_awkward_station_sizes(piles, types):
    """the two-slots-per-lane pass lays a wave's units end to end over its 128 virtual lanes (floor(128 / S) units per wave, one of them may
    straddle the two virtual waves) and the walk files its cars env by env: where the boundaries fall depends on the pile count.  The same
    program as above at 777 envs (the last wave of each station partly filled) on station sizes of 8 .. 64 piles, both kinds in both places."""
    kw = dict(HUB_KW, station_list=list(piles), station_type_list=list(types), fcev_permeate=0.03)
    ref_calls, ref_end = compat_program(777, "wave", seed=77, kw=kw)
    for form in ("packed", "packed_own_walks"):
        calls, end = compat_program(777, form, seed=77, kw=kw)
        first = [k for k, (a, b) in enumerate(zip(ref_calls, calls)) if a != b]
        assert len(calls) == len(ref_calls) and not first, (piles, form, "first differing call", first[0])
        for k, (a, b) in enumerate(zip(ref_end, end)):
            assert np.array_equal(a, b, equal_nan=True), (piles, form, "end state array", k)


def philox_program(n, kw, seed, **opts):
    """a day and a bit on the device-pointer path (resets at the episode boundaries), six steps as a graph replay, a masked reset and
    masked steps -> (per-sample digests, end state arrays)"""
    chub = hub()
    from charginghub_env_amd import multi_gpu
    from charginghub_env_amd._lib import check
    v = chub.VecChargingHub(n, seed=seed, fused_step="off", **opts, **kw)
    lib, h = v._lib, v._h
    st = multi_gpu.Stream(0)
    acts = [multi_gpu.DeviceBuffer(n * v.act_dim * 4) for _ in range(4)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, 41, b, st.ptr)
    D = v.obs_dim
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]
    obs, rew, done = multi_gpu.DeviceBuffer(n * D * 4), multi_gpu.DeviceBuffer(n * 4), multi_gpu.DeviceBuffer(n)
    calls = []
    for i in range(100):
        if i % 96 == 0:
            v.reset_device(obs.ptr, stream=st.ptr)
            calls.append(digest(obs.to_host(np.float32, (n, D), st.ptr)))
        v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)
        if i % 7 == 0 or i >= 94:
            calls.append(digest(packed[i & 1].to_host(np.float32, (n, D + 2), st.ptr)))
    st.sync()
    v.graph_begin(st.ptr)
    for i in range(6):
        v.step_device_packed(acts[i % 4].ptr, packed[i & 1].ptr, stream=st.ptr)
    g = v.graph_end(st.ptr)
    v.graph_launch(g, st.ptr)
    calls.append(digest(packed[1].to_host(np.float32, (n, D + 2), st.ptr)))
    v.graph_destroy(g)
    rs = np.random.RandomState(3)
    for k in range(4):  # per-env clocks: a masked reset between masked steps
        m = np.ascontiguousarray(rs.uniform(size=n) < 0.4, dtype=np.uint8)
        if k == 1:
            check(lib.chub_reset_envs_device(h, m.ctypes.data, None, None, obs.ptr, st.ptr))
        else:
            check(lib.chub_step_envs_device(h, m.ctypes.data, acts[k % 4].ptr, None, obs.ptr, rew.ptr, done.ptr, st.ptr))
        calls.append(digest(obs.to_host(np.float32, (n, D), st.ptr)))
    end = [digest(*v.slots()), digest(v.station_scalars()), digest(v.env_clocks())]
    flags = (v.uses_packed_kernel, v.uses_xcd_order)
    v.close()
    st.destroy()
    for b in acts + packed + [obs, rew, done]:
        b.free()
    return calls + end, flags


C4_KW = dict(HUB_KW, renew_fluctuate=0.2, price_fluctuate=0.1)
C5_KW = dict(HUB_KW, station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3)


@pytest.mark.parametrize("label,n,kw", [("c4", 65536, C4_KW), ("c5", 262144, C5_KW)])
def test_philox_work_orders_and_tiles_are_bit_identical_at_bench_size(label, n, kw):
    """chub_options.work_order x chub_options.tile at the headline size (65 536 x [20,25]: 5958 tiles, the XCD-aware order is its default) and
    at BASELINE configs[4] (262 144 x [32,32]: 8192 tiles of 512 x 4, the dispatcher's order is its default): the same tiles in another order
    and the same slots on another tile -- every sampled packed output, the graph replay, the masked calls, slot state and station records."""
    base, flags = philox_program(n, kw, 9, work_order="dispatch", tile="small")
    assert flags == (True, False)
    for order, tile in (("auto", "small"), ("dispatch", "large"), ("auto", "large")):
        got, f = philox_program(n, kw, 9, work_order=order, tile=tile)
        assert f[0]
        diff = [k for k, (a, b) in enumerate(zip(base, got)) if a != b]
        assert len(got) == len(base) and not diff, (label, order, tile, "first differing sample", diff[0])
    if label == "c4":  # (the default handle of the headline size IS the XCD-aware order on the small tile)
        _, f = philox_program(4096, kw, 9)
        assert f == (True, True)
