"""The production (PHILOX) slot kernel against the reference's recorded trajectories, directly.

The production streams are this build's own definition, so k_slot_packed cannot be compared with a reference run seed for
seed.  Tape mode (include/chub.h) feeds the SAME kernel what the reference's streams decided -- per station and step the
queue / arrival / admission decisions, per admitted car its arrival SoC, target SoC and stay -- all derived here from
the golden fixtures (tests/golden/env_*.npz, recorded from the unmodified reference): the per-slot state the kernel then
produces must be the reference's bit for bit, the station counts exactly, and the station power sums (computed through
the production 2^-19 kW integer path) within 1e-5 relative (+ 1e-6 kW) of the reference's sequential f32 sums.  All 18 fixtures;
the resets run the same way (chub_reset_tape: evs_reset through k_slot_packed<.., RESET, ..>).
"""
import numpy as np
import pytest

import orclib
from orclib import orc

pytestmark = pytest.mark.gpu

FAST, SLOW = 0, 1


def _levels():
    """target SoC of level k (RandomUtil::uniform_rand(80, 100) at rand() % 1000 == k, CHS.hpp:35-44) -> k"""
    return {np.float32(orc.orc_uniform_level(k, 80.0, 100.0)).tobytes(): k for k in range(1000)}


def _soc_to_time(typ, soc, cp):
    f = orc.orc_curve_fast if typ == FAST else orc.orc_curve_slow
    return np.float32(f(2, float(soc), int(cp)))


def _new_car(typ, cp, init_soc, target_soc, stay, levels):
    """what add_car drew for a recorded new car (CHS.hpp:864-877 / 1029-1042): arrival SoC, target level, extra stay"""
    lev = levels[np.float32(target_soc).tobytes()]
    need = np.float32(_soc_to_time(typ, target_soc, cp) - _soc_to_time(typ, init_soc, cp))
    late = int(stay) - int(np.ceil(need))
    assert 0 <= late < 16, (late, need, stay)
    return lev, late


def _pk_word(n, prev_slots, cur_slots, line_before, line_after, flow):
    """the packed station-level decisions (64-bit layout of include/chub.h, tape mode) that make receive_car
    (CHS.hpp:1272-1316 / 1583-1627) come out as recorded: renege survivors, arrivals, balk survivors.  n = 0: a station
    without piles still queues, reneges and balks; nobody is ever admitted."""
    left = prev_slots[7] - prev_slots[8]                       # stay_time - already_stay_time
    stays = (prev_slots[0] > 0.5) & (left > 1)                  # cars still there after remove_car
    empties = n - int(stays.sum())
    new = (cur_slots[0] > 0.5) & (cur_slots[8] == 0)            # admitted this step
    assign = int(new.sum())
    flow = int(flow)
    if assign < empties:                                        # everybody queued or arriving got a slot
        line_r = assign - flow
    elif line_after < 10:
        line_r = line_after + assign - flow
    else:                                                       # queue saturated at max_line: any consistent count will do
        line_r = line_before
    assert 0 <= line_r <= line_before, (line_r, line_before, assign, flow, empties, line_after)
    assert min(line_r + flow, empties) == assign and min(line_r + flow - assign, 10) == line_after
    assert 0 <= flow <= 9
    pk = (1 << line_r) - 1                                      # the first line_r queued cars stay (CHS.hpp:1286-1293)
    pk |= flow << 10                                            # arrivals (the fast station records this count, CHS.hpp:1617)
    for l in range(11):                                         # balk survivors per queue length (the slow station's count, CHS.hpp:1306)
        pk |= flow << (14 + 4 * l)
    return pk, new


ALL_FIXTURES = ["env_c1_envtest", "env_c3_random", "env_c2_random", "env_c5_random", "env_slow_only_fcev", "env_clamp", "env_full_tank",
                "env_fcev_queue", "env_constant", "env_small_fast_neg", "env_fcev_queue_deep", "env_big_100_70",
                "env_slow_slow", "env_fast_fast", "env_no_electrolyser", "env_permeate_cap", "env_one_pile", "env_constant_swapped"]
assert sorted(ALL_FIXTURES) == sorted(orclib.GOLDEN_ENV)


@pytest.mark.parametrize("name", ["env_c5_random", "env_big_100_70", "env_small_fast_neg"])
def test_large_tile_replays_reference_fixture(name):
    """the second workgroup tile of the packed kernel (512 lanes x 4 slots: what handles of 10 M slots and more run) against the
    reference directly: the same replay on a handle forced onto it"""
    test_packed_kernel_replays_reference_fixture(name, tile="large")


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_packed_kernel_replays_reference_fixture(name, tile="small"):
    """Every reference fixture through the production kernel, evs_reset included: the k_slot_packed instantiations for steps and
    resets, stations of up to 64 piles and beyond (BIG: env_big_100_70), a station without piles (env_c2_random,
    env_slow_only_fcev), a 3-pile hub whose reset records a negative flow_in (env_small_fast_neg)."""
    import charginghub_env_amd as chub
    g = orclib.load_golden(name)
    piles = [int(x) for x in g["kw_station_list"]]
    types = [int(x) for x in g["kw_station_type"]]
    cp = bool(g["kw_constant_charging"])
    kw = dict(station_list=piles, station_type_list=["fast" if t == 0 else "slow" for t in types], constant_charging=cp,
              hydro_prod_rate=float(g["kw_hydro_prod_rate"]), hydro_store_vlt=float(g["kw_hydro_store_vlt"]),
              init_soc=float(g["kw_init_soc"]), fc_max_power=float(g["kw_fc_max_power"]),
              fcev_permeate=float(g["kw_fcev_permeate"]))
    n_envs = 3                                                   # every env replays the same tape
    v = chub.VecChargingHub(n_envs, seed=1, rng="philox", slot_kernel="packed", tile=tile, fused_step="off" if tile == "large" else "auto", **kw)
    assert v.uses_packed_kernel
    S0, S1 = piles
    S = S0 + S1
    levels = _levels()
    # every arrival SoC the reference drew becomes a class of the class table
    socs = set()
    for key in ("reset_slots0", "reset_slots1", "slots0", "slots1"):
        a = g[key]
        occ = a[:, 0, :] > 0.5
        socs.update(np.unique(a[:, 5, :][occ]).tolist())
    socs = np.array(sorted(socs), dtype=np.float32)
    per_episode = name == "env_c5_random"                       # one fixture registers its classes episode by episode (chub_tape_clear_soc)
    cls_of = {}
    if not per_episode:
        ids = v.tape_register_soc(socs)
        cls_of = {np.float32(s).tobytes(): int(i) for s, i in zip(socs, ids)}
    rep = lambda a: np.repeat(np.asarray(a)[None], n_envs, axis=0)
    steps = int(g["steps_per_episode"])

    def compare(cur, st, what):
        sl = v.slots()
        sc = v.station_scalars()
        for e in range(n_envs):
            for k in (0, 1):
                got, want = sl[k][e], cur[k]
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, what, k, got, want)
                ref = st[6 * k:6 * k + 6]
                assert np.array_equal(sc[e, k, 3:6], ref[3:6]), (name, what, k, sc[e, k], ref)          # car_number, line, flow_in
                # the production sums: every car's power to the nearest 2^-19 kW, added as integers -- within the north star's 1e-5
                # relative of the reference's sequential f32 sums
                assert np.allclose(sc[e, k, :3], ref[:3], rtol=1e-5, atol=1e-6), (name, what, k, sc[e, k], ref)

    i = 0
    n_new = 0
    for ep in range(int(g["episodes"])):
        # ---- evs_reset (CHS.hpp:1209-1231 / 1520-1542) fed what the reference drew: the unit's initial occupancy and, per car it
        # put into a slot, arrival SoC, target level and extra stay
        prev = [g["reset_slots0"][ep], g["reset_slots1"][ep]]
        rst = g["reset_stations"][ep]
        if per_episode:                                          # evs_reset wipes every slot: the old classes are free again
            mine = set()
            for a in (prev[0][None], prev[1][None], g["slots0"][ep * steps:(ep + 1) * steps], g["slots1"][ep * steps:(ep + 1) * steps]):
                mine.update(np.unique(a[:, 5, :][a[:, 0, :] > 0.5]).tolist())
            mine = np.array(sorted(mine), dtype=np.float32)
            v.tape_clear_soc()
            cls_of = {np.float32(s).tobytes(): int(i) for s, i in zip(mine, v.tape_register_soc(mine))}
            assert max(cls_of.values()) == len(mine) - 1 < len(socs)
        occ = np.zeros((2, n_envs), dtype=np.uint32)
        car = np.zeros((S, 2), dtype=np.uint32)
        for k, off, n in ((0, 0, S0), (1, S0, S1)):
            flow = int(rst[6 * k + 5])                          # flow_in_number[-1]: the fast station's raw draw, the slow one's survivors
            occ[k, :] = (flow & 0xFFFF) | (max(flow, 0) << 16)
            for s in np.nonzero(prev[k][0] > 0.5)[0]:
                lev, late = _new_car(types[k], cp, prev[k][5, s], prev[k][6, s], prev[k][7, s], levels)
                car[off + s] = [cls_of[np.float32(prev[k][5, s]).tobytes()], lev | (late << 16)]
                n_new += 1
        if name == "env_constant" and ep == 0:
            # (the other way in, kept covered: a Philox reset overwritten through chub_set_slots / chub_set_station_queue)
            v.reset()
            rows = np.full((S, 6), -1, dtype=np.int32)
            for k, off, n in ((0, 0, S0), (1, S0, S1)):
                for s in np.nonzero(prev[k][0] > 0.5)[0]:
                    rows[off + s] = [cls_of[np.float32(prev[k][5, s]).tobytes()], levels[np.float32(prev[k][6, s]).tobytes()],
                                     int(prev[k][7, s]), int(prev[k][8, s]), 0, int(prev[k][1, s] > 0.5)]
            v.set_slots(rep(rows))
            v.set_station_queue(rep([int(rst[4]), int(rst[10])]))
        else:
            v.reset_tape(occ, rep(car))
            compare(prev, rst, ("reset", ep))
        line = [int(rst[4]), int(rst[10])]
        for t in range(steps):
            cur = [g["slots0"][i], g["slots1"][i]]
            st = g["stations"][i]
            pk = np.zeros((2, n_envs), dtype=np.uint64)
            car = np.zeros((S, 2), dtype=np.uint32)
            for k, off, n in ((0, 0, S0), (1, S0, S1)):
                line_after, flow = int(st[6 * k + 4]), int(st[6 * k + 5])
                word, new = _pk_word(n, prev[k], cur[k], line[k], line_after, flow)
                pk[k, :] = word
                for s in np.nonzero(new)[0]:
                    lev, late = _new_car(types[k], cp, cur[k][5, s], cur[k][6, s], cur[k][7, s], levels)
                    car[off + s] = [cls_of[np.float32(cur[k][5, s]).tobytes()], lev | (late << 16)]
                    n_new += 1
                line[k] = line_after
            v.step_tape(rep(g["action"][i]), pk, rep(car))
            compare(cur, st, (ep, t))
            prev = cur
            i += 1
    assert n_new > 40                                            # the tape really admitted cars
    v.close()
