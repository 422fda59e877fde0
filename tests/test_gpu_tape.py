"""The production (PHILOX) slot kernel against the reference's recorded trajectories, directly.

The production streams are this build's own definition, so k_slot_packed cannot be compared with a reference run seed for
seed.  Tape mode (include/chub.h) feeds the SAME kernel what the reference's streams decided -- per station and step the
queue / arrival / admission decisions, per admitted car its arrival SoC, target SoC and stay -- all derived here from
the golden fixtures (tests/golden/env_*.npz, recorded from the unmodified reference): the per-slot state the kernel then
produces must be the reference's bit for bit, the station counts exactly, and the station power sums (computed through
the production 2^-19 kW integer path) within 1e-5 relative (+ 1e-6 kW) of the reference's sequential f32 sums.  All 18 fixtures;
the resets run the same way (chub_reset_tape: evs_reset through k_slot_packed<.., RESET, ..>).

Round 5: the tape goes through the per-env TAIL as well (chub_step_tape_env / chub_reset_tape_env): the exogenous normals numpy drew for
the reference, renew_reset's days, the forecourt's arrivals and their SoCs, and hy_power_speed_list as the reference's constructor built it,
all from the fixture -- so observation, reward, done and every telemetry column of k_env<.., PHILOX> (and of the tail half of the one-launch
step k_step_fused) are held to the reference's recorded values directly, not through the oracle.  Columns that do not depend on the station
power sums must agree to 1e-9; those that do inherit the production sums' distance from the reference's sequential f32 sums (the north
star's 1e-5 relative; the integer sums are the more exact of the two).
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
                "env_slow_slow", "env_fast_fast", "env_no_electrolyser", "env_permeate_cap", "env_one_pile", "env_constant_swapped",
                "env_past_done", "env_past_done_c2", "env_defaults", "env_tank_floor", "env_tank_brim"]
assert sorted(ALL_FIXTURES) == sorted(orclib.GOLDEN_ENV)


TIGHT = 1e-9
RTOL = 1e-5  # north_star: "within 1e-5 relative for battery SoC, power and reward floats"
# (CHUB_TAPE_TOL scales the bars of the columns downstream of the station power sums: how DESIGN.md's "observed" figures were found)
TOL = float(__import__("os").environ.get("CHUB_TAPE_TOL", "1"))
# telemetry columns (charginghub-env_amd/_lib.py: TELEMETRY_NAMES) that no station power sum reaches: the exogenous series, the price, the
# forecourt's demand.  (The hydrogen system is not among them: the electrolyser clamp MGR:160-180 and the fuel cell's draw on the tank,
# capped by the EV load HYD:409-430, both look at the station sums.)
TEL_EXO = [5, 16, 17, 18]               # total_mass_need, re_pv_power, re_wd_power, price_next
# ---- the absolute part of the bars: not a blanket floor, but what the arithmetic needs.  The one place where the production
# path may differ from the reference's arithmetic at all is a station's three power sums: the reference adds its S slot powers one by one in
# f32 (CHS.hpp:1244-1255), so ITS sum is off the exact one by up to (S - 1) * 2^-24 * sum|p| (<= max_power); the production sum is exact on a
# 2^-19 kW grid, off by up to S * 2^-20 kW.  `slack` below is that distance per station, in kW, from the reference's own record of the
# step; every column downstream gets slack times the factor by which the reference's formulas pass a kW of station power on
# (MGR:233-269, HYD:409-430: 1500 / 119.6 g of hydrogen per kW of fuel cell; incomes at 0.42 / 4 + price / 4 $ per kW; reward = income / 50;
# MGR:399-402: a station column of the observation is (P - half) / half).  Next to it the relative 1e-5 of the north star.
G_PER_KW = 1500 / 119.6
TEL_KW = [8, 10, 11, 12, 13]            # fc_power, re_used_renew, re_ev_power_0 / _1, re_hydrogen_power
TEL_GRAM = [4, 6, 7, 9]                 # capacity, hy_use, not_meet, hy_to_use
TEL_CLAMP = [0, 1, 2]                   # hy_act, hy_flow_speed, all_power_second: functions of the clamp's table index alone -- exact unless it flips
OBSERVED = {}                           # (kind, column) -> largest |got - want| / bar seen over the whole session (reported by the last test)


def _slack(st_row, piles):
    """per station, kW: how far the reference's sequential f32 sums of this step and the production's integer sums may lie apart"""
    return [max(piles[k] - 1, 0) * 2.0 ** -24 * float(st_row[6 * k + 2]) + piles[k] * 2.0 ** -20 for k in (0, 1)]


def _hold(kind, cols, got, want, atol, ctx):
    got, want, atol = np.atleast_1d(got).astype(np.float64), np.atleast_1d(want).astype(np.float64), np.broadcast_to(np.atleast_1d(atol), np.atleast_1d(want).shape)
    bar = RTOL * TOL * np.abs(want) + atol * TOL
    err = np.abs(got - want)
    for c, e_, b_ in zip(cols, err, bar):
        OBSERVED[(kind, int(c))] = max(OBSERVED.get((kind, int(c)), 0.0), float(e_ / b_) if b_ > 0 else (0.0 if e_ == 0 else np.inf))
    assert (err <= bar).all(), ctx + (kind, list(cols), got, want, bar)


def _check_tail(v, g, i, name, n_envs, what, piles):
    """observation / reward / done / telemetry of the production tail against the reference's record of step i"""
    o64, r64, tel = v.obs_f64(), v.reward_f64(), v.telemetry()
    sc = v.station_scalars()
    D = o64.shape[1]
    want_o, want_t = g["obs"][i], g["telem"][i]
    slack = _slack(g["stations"][i], piles)
    dP = slack[0] + slack[1]
    price = max(float(np.abs(g["real_state"][:, 1]).max()), float(np.abs(g["reset_real_state"][:, 1]).max()))  # (the tariff + its noise: the fixture's largest)
    k_money = 0.42 / 4 + price / 4 + 6 / 1000 * G_PER_KW   # $ per kW: charging income, the grid's price, the fuel cell's hydrogen
    cap_mass = float(want_t[4]) / float(want_t[3]) if want_t[3] > 0 else np.inf
    # observation layout (MGR:364-373): [sin t, price, {min, charge, max, line / 5} per station with piles, H2 SOC, pv, wd]
    exact_cols = [0, 1, D - 2, D - 1] + [c for c in range(2, D - 3) if (c - 2) % 4 == 3]
    with_piles = [k for k in (0, 1) if piles[k] > 0]
    for e in range(n_envs):
        ctx = (name, what, e)
        assert np.allclose(o64[e, exact_cols], want_o[exact_cols], rtol=TIGHT, atol=TIGHT), ctx + ("obs (exogenous, price, queues)", o64[e], want_o)
        for j, k in enumerate(with_piles):   # (P - half) / half with half = transformer_limit / 2 (MGR:399-402)
            cols = [2 + 4 * j, 3 + 4 * j, 4 + 4 * j]
            _hold("obs_station", cols, o64[e, cols], want_o[cols], slack[k] / (float(sc[e, k, 7]) / 2), ctx)
        _hold("obs_soc", [D - 3], o64[e, D - 3], want_o[D - 3], G_PER_KW * dP / cap_mass, ctx)
        _hold("reward", [0], r64[e], g["reward"][i], k_money * dP / 50, ctx)
        assert np.array_equal(tel[e, 19:22], want_t[19:22]), ctx + ("fcev ints", tel[e, 19:22], want_t[19:22])
        assert np.allclose(tel[e, TEL_EXO], want_t[TEL_EXO], rtol=TIGHT, atol=TIGHT), ctx + ("telemetry (exogenous)", tel[e], want_t)
        _hold("tel_clamp", TEL_CLAMP, tel[e, TEL_CLAMP], want_t[TEL_CLAMP], 0.0, ctx)
        _hold("tel_kw", TEL_KW, tel[e, TEL_KW], want_t[TEL_KW], dP, ctx)
        _hold("tel_gram", TEL_GRAM, tel[e, TEL_GRAM], want_t[TEL_GRAM], G_PER_KW * dP, ctx)
        _hold("tel_soc", [3], tel[e, 3], want_t[3], G_PER_KW * dP / cap_mass, ctx)
        _hold("tel_money", [14], tel[e, 14], want_t[14], k_money * dP, ctx)
        _hold("tel_reward", [15], tel[e, 15], want_t[15], k_money * dP / 50, ctx)


@pytest.mark.parametrize("name", ["env_c5_random", "env_big_100_70", "env_small_fast_neg"])
def test_large_tile_replays_reference_fixture(name):
    """the second workgroup tile of the packed kernel (512 lanes x 4 slots: what handles of 10 M slots and more run) against the
    reference directly: the same replay on a handle forced onto it"""
    test_packed_kernel_replays_reference_fixture(name, tile="large")


@pytest.mark.parametrize("name", [n for n in ALL_FIXTURES if n != "env_big_100_70"])  # (stations of more than 64 piles never take the one-launch form)
def test_single_launch_step_replays_reference_fixture(name):
    """the same replay with the steps as ONE launch (k_step_fused<.., TAPE>: slot body, station records and the tails of the workgroup's
    envs in one kernel) -- what every PHILOX batch of up to 384 slot workgroups runs"""
    test_packed_kernel_replays_reference_fixture(name, fused="on")


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_packed_kernel_replays_reference_fixture(name, tile="small", fused="off"):
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
              fcev_permeate=float(g["kw_fcev_permeate"]), renew_fluctuate=float(g["kw_renew_fluctuate"]),
              price_fluctuate=float(g["kw_price_fluctuate"]), hydro_loss=float(g["kw_hydro_loss"]))
    n_envs = 3                                                   # every env replays the same tape
    v = chub.VecChargingHub(n_envs, seed=1, rng="philox", slot_kernel="packed", tile=tile, fused_step=fused, **kw)
    assert v.uses_packed_kernel and v.uses_fused_step == (fused == "on")
    v.set_telemetry(True)
    v.set_hy_table(g["hy_table"])                                # hy_power_speed_list as the reference's constructor built it (HYD:154-157)
    hv_w = 1 + g["hv_soc"].shape[1]
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
    S_ = sum(piles)
    # the reference's constructor runs one reset() (MGR:120): its exogenous draws shape the OU states the first episode starts from
    v.reset_tape(np.zeros((2, n_envs), dtype=np.uint32), np.zeros((n_envs, S_, 2), dtype=np.uint32), rep(g["ctor_days"]), rep(g["ctor_z"]))

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
                # relative of the reference's sequential f32 sums, + the distance the two kinds of sum may lie apart (_slack)
                _hold("station_sums", [3 * k, 3 * k + 1, 3 * k + 2], sc[e, k, :3], ref[:3], _slack(st, piles)[k], (name, what, e))

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
            # (the other way in, kept covered: the slots of a reset overwritten through chub_set_slots / chub_set_station_queue; the tail's
            # side of the reset -- days, OU states, tank -- from the tape all the same, on an empty hub)
            v.reset_tape(np.zeros((2, n_envs), dtype=np.uint32), np.zeros((n_envs, S, 2), dtype=np.uint32), rep(g["reset_days"][ep]), rep(g["reset_z"][ep]))
            rows = np.full((S, 6), -1, dtype=np.int32)
            for k, off, n in ((0, 0, S0), (1, S0, S1)):
                for s in np.nonzero(prev[k][0] > 0.5)[0]:
                    rows[off + s] = [cls_of[np.float32(prev[k][5, s]).tobytes()], levels[np.float32(prev[k][6, s]).tobytes()],
                                     int(prev[k][7, s]), int(prev[k][8, s]), 0, int(prev[k][1, s] > 0.5)]
            v.set_slots(rep(rows))
            v.set_station_queue(rep([int(rst[4]), int(rst[10])]))
        else:
            v.reset_tape(occ, rep(car), rep(g["reset_days"][ep]), rep(g["reset_z"][ep]))
            compare(prev, rst, ("reset", ep))
        o64 = v.obs_f64()
        D_ = o64.shape[1]
        cols = [0, 1, D_ - 3, D_ - 2, D_ - 1] + [c for c in range(2, D_ - 3) if (c - 2) % 4 == 3]
        via_set_slots = name == "env_constant" and ep == 0     # (the reset there ran on an empty hub: only the columns no station reaches)
        if via_set_slots:
            cols = [0, 1, D_ - 3, D_ - 2, D_ - 1]
        for e in range(n_envs):
            assert np.allclose(o64[e, cols], g["reset_obs"][ep][cols], rtol=TIGHT, atol=TIGHT), (name, "reset obs", ep, o64[e], g["reset_obs"][ep])
            if not via_set_slots:  # (the station columns: the reset's own sums, held as a step's are)
                sl_ = _slack(rst, piles)
                sc_ = v.station_scalars()
                for j, k in enumerate([k for k in (0, 1) if piles[k] > 0]):
                    cs = [2 + 4 * j, 3 + 4 * j, 4 + 4 * j]
                    _hold("reset_obs_station", cs, o64[e, cs], g["reset_obs"][ep][cs], sl_[k] / (float(sc_[e, k, 7]) / 2), (name, "reset obs", ep))
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
            hv = np.zeros(hv_w, dtype=np.uint32)
            hv[0] = int(g["telem"][i][19])                      # hvs.arrive_number
            hv[1:1 + hv[0]] = g["hv_soc"][i][:hv[0]].view(np.uint32)
            _, _, done = v.step_tape(rep(g["action"][i]), pk, rep(car), rep(g["exo_z"][i]), rep(hv))[:3]
            compare(cur, st, (ep, t))
            assert all(bool(d) == bool(g["done"][i]) for d in done)
            _check_tail(v, g, i, name, n_envs, (ep, t), piles)
            prev = cur
            i += 1
    assert n_new > 40                                            # the tape really admitted cars
    v.close()


def test_tape_handles_refuse_what_would_mix_the_two_sets_of_classes():
    """a handle whose class rows hold the caller's arrival SoCs admits cars through the tape only; cleared classes need a reset first;
    SoCs no stay of which fits the state word are refused"""
    import charginghub_env_amd as chub
    kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2,
              fc_max_power=100.0, fcev_permeate=0.01)
    n = 2
    v = chub.VecChargingHub(n, seed=1, rng="philox", slot_kernel="packed", **kw)
    v.reset()                                                    # (a production handle until classes are registered)
    v.step(np.zeros((n, v.act_dim), dtype=np.float32))
    with pytest.raises(chub.ChubError):
        v.tape_register_soc(np.array([50.0, -5.0], dtype=np.float32))
    v.step(np.zeros((n, v.act_dim), dtype=np.float32))           # the refused batch changed nothing
    ids = v.tape_register_soc(np.array([30.0, 50.0], dtype=np.float32))
    assert list(ids) == [0, 1]
    for call in (lambda: v.reset(), lambda: v.step(np.zeros((n, v.act_dim), dtype=np.float32)),
                 lambda: v.step_bits(*v.pack_actions(np.zeros((n, v.act_dim), dtype=np.float32)))):
        with pytest.raises(chub.ChubError):
            call()
    S = v.n_slots
    occ, car, pk = np.zeros((2, n), dtype=np.uint32), np.zeros((n, S, 2), dtype=np.uint32), np.zeros((2, n), dtype=np.uint64)
    v.reset_tape(occ, car)
    v.step_tape(np.zeros((n, v.act_dim), dtype=np.float32), pk, car)
    v.tape_clear_soc()
    with pytest.raises(chub.ChubError):
        v.step_tape(np.zeros((n, v.act_dim), dtype=np.float32), pk, car)
    v.tape_register_soc(np.array([40.0], dtype=np.float32))
    v.reset_tape(occ, car)
    v.step_tape(np.zeros((n, v.act_dim), dtype=np.float32), pk, car)
    v.close()


def test_zz_report_the_observed_distances():
    """(runs last in this file) what the replays above observed, per kind of column: the largest |got - want| as a fraction of its bar
    (rtol 1e-5 + the slack of the reference's own f32 sums passed through the reference's formulas) -- printed with -s and left in
    gpurun_out/tape_observed.json for DESIGN.md section 2"""
    import json
    import os
    if not OBSERVED:
        pytest.skip("no replay ran in this session")
    worst = {}
    for (kind, c), r in OBSERVED.items():
        worst[kind] = max(worst.get(kind, 0.0), r)
    print("\nlargest error / bar per kind of column:", json.dumps(worst, indent=1, sort_keys=True))
    assert max(worst.values()) <= 1.0
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump({"tol_scale": TOL, "rtol": RTOL, "worst_error_over_bar": worst,
                   "per_column": {"%s[%d]" % k: v for k, v in sorted(OBSERVED.items())}}, open(os.path.join(out, "tape_observed.json"), "w"), indent=1)
