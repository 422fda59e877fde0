"""Oracle (CPU restatement) vs golden trajectories recorded from the unmodified reference
(tests/golden/env_*.npz, generator oracle/gen/gen_env_golden.py).  No GPU, no reference tree needed."""
import numpy as np
import pytest

import orclib
from orclib import OrcEnv

RTOL = 1e-12  # f64 host arithmetic restated op-for-op; observed agreement is ~1e-15


def _close(a, b, what, rtol=RTOL, atol=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.allclose(a, b, rtol=rtol, atol=atol), (what, a, b, np.abs(a - b).max())


def replay(name, make_env_and_check):
    g = orclib.load_golden(name)
    cfg = orclib.golden_config(g)
    return g, cfg


@pytest.mark.parametrize("name", orclib.GOLDEN_ENV)
def test_env_trajectory_matches_reference(name):
    _check_trajectory(name)


@pytest.mark.parametrize("name", orclib.GOLDEN_ENV_BIG)
def test_env_trajectory_matches_reference_on_stations_of_more_than_256_piles(name):
    """The same check on liboracle_big.so (the same source with room for 4096 piles per station)."""
    with orclib.big_oracle():
        _check_trajectory(name)


def _check_trajectory(name):
    g = orclib.load_golden(name)
    cfg = orclib.golden_config(g)
    steps = int(g["steps_per_episode"])
    seeds = {int(ep): (int(a), int(b)) for ep, a, b in g["seeds"]} if g["seeds"].size else {}
    # The reference's constructor consumes its two C++ streams in a fixed order: the station constructors' evs_reset,
    # then the 101-step electrolyser sweep with live FCEV demand (HYD:154-157), whose result hy_power_speed_list depends on
    # those draws whenever a tank clamp binds.  Every fixture records the seeds in force at construction (env_c1_envtest:
    # the process defaults, rand() never srand()ed -> 1, e -> 1), so the table is reproduced, not injected.
    env = OrcEnv(cfg, ctor_seeds=tuple(int(x) for x in g["ctor_seeds"]))
    assert np.array_equal(env.hy_table(), g["hy_table"]), name
    env.reset(g["ctor_days"], g["ctor_z"])  # MGR:120, shapes the persistent OU states
    S0 = cfg.piles[0]
    i = 0
    ret = 0.0
    draw = 0.0
    for ep in range(int(g["episodes"])):
        if ep in seeds:
            env.seed_compat(*seeds[ep])
        obs = env.reset(g["reset_days"][ep], g["reset_z"][ep])
        _close(obs, g["reset_obs"][ep], ("reset_obs", ep))
        st = np.concatenate([env.station_scalars(0)[:6], env.station_scalars(1)[:6]])
        assert np.array_equal(st, g["reset_stations"][ep]), ("reset stations", ep)
        for t in range(steps):
            obs, r, d = env.step(g["action"][i], g["exo_z"][i])
            # integer / index quantities: exact
            st = np.concatenate([env.station_scalars(0)[:6], env.station_scalars(1)[:6]])
            assert np.array_equal(st, g["stations"][i]), (name, ep, t, st, g["stations"][i])
            assert np.array_equal(env.station_slots(0).view(np.uint32), g["slots0"][i].view(np.uint32)), (ep, t)
            assert np.array_equal(env.station_slots(1).view(np.uint32), g["slots1"][i].view(np.uint32)), (ep, t)
            assert d == bool(g["done"][i])
            tel = env.telemetry()
            assert tel[19] == g["telem"][i][19] and tel[20] == g["telem"][i][20] and tel[21] == g["telem"][i][21], (name, ep, t)
            assert env.q_overflow() == 0
            # floats
            _close(obs, g["obs"][i], (name, "obs", ep, t))
            _close(r, g["reward"][i], (name, "reward", ep, t))
            _close(tel[:19], g["telem"][i][:19], (name, "telem", ep, t), rtol=1e-11, atol=1e-9)
            # what the reference class carries after the step beyond that (recorded as `attrs`): ev_power_list / ev_power_sum after
            # the fuel cell through the attributes made of them, and real_state
            at = dict(zip([str(x) for x in g["attr_names"]], g["attrs"][i]))
            rpd = tel[27] / 4
            _close([tel[26], rpd, -rpd * tel[24], -rpd * tel[25], 0.42 / 4 * tel[29], 0.21 / 4 * tel[34], 0.8 * (tel[32] + tel[37])],
                   [at["real_charging_power"], at["re_price_dollar"], at["re_income_evs_cost_list_0"], at["re_income_evs_cost_list_1"],
                    at["re_income_evs_list_0"], at["re_income_evs_list_1"], at["re_income_evs_serve"]], (name, "attrs", ep, t), rtol=1e-11, atol=1e-9)
            draw = (draw if t else 0.0) + (tel[24] + tel[25] + tel[13])
            _close(draw, at["cumulated_draw_ele"], (name, "cumulated_draw_ele", ep, t), rtol=1e-11, atol=1e-9)
            cols = [18] + [c for k in (0, 1) if int(g["kw_station_list"][k]) > 0 for c in range(28 + 5 * k, 32 + 5 * k)] + [3, 16, 17]
            _close(tel[cols], g["real_state"][i][1:], (name, "real_state", ep, t), rtol=1e-11, atol=1e-9)
            if ep == 0:
                ret += r
            i += 1
    if name == "env_c1_envtest":
        # known answer recorded by the survey: return 34.858789741560585, final SOC 0.1525
        assert abs(ret - 34.858789741560585) < 1e-9
        assert abs(env.telemetry()[3] - 0.1525) < 1e-12
