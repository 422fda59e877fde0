#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/env_*.npz from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and oracle/_ref/libchs_ref.so).  The
reference's Python host (evcssp_manager.py, Aggregator_Simple.py, hydro_sys.py, renewable.py) is
imported as it lies; the two module names it needs that the image lacks are bound by
oracle/gen/stubs/: `pyevstation` (ctypes binding onto the real CHS.hpp build) and `gym` (inert base
class / Box / seeding names).  All arithmetic runs in reference code.

Each fixture is DATA: constructor kwargs, the two C++ stream seeds installed right before reset(),
the per-step action tape, the exogenous tape (pv_day, wd_day and every np.random.normal() draw the
reference made, in order) and the resulting obs / reward / done / telemetry / station outputs.
"""
import io
import contextlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CHUB_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pyevstation  # noqa: E402  (the stub binding; registers the real C++ core)

sys.modules["lion_cpp20.pyevstation"] = pyevstation
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    from evcssp_env_cpp.envs.evcssp_manager import EvcsspManagerEnv_v6  # noqa: E402
import orclib  # noqa: E402

GOLD = os.environ.get("CHUB_GOLD_OUT", os.path.join(ROOT, "tests", "golden"))  # (a scratch directory: compare before replacing fixtures)

TELEM = ["hy_act", "hy_flow_speed", "all_power_second", "Store_SOC", "capacity", "total_mass_need", "hy_use",
         "not_meet", "fc_power", "hy_to_use", "re_used_renew", "ev0", "ev1", "re_hydrogen_power", "income",
         "reward", "re_pv_power", "re_wd_power", "price_next", "arrive_number", "hvs_line", "queue_len"]


# attributes of the reference class that trainers and evaluation scripts read after step() (MGR:128-130, 229-297, 372): recorded
# per step so that the drop-in class can be held to them
ATTRS = ["cumulated_draw_ele", "cumulated_income", "acumulate_reward", "deviation", "test_penalty", "penalty", "re_price_dollar",
         "real_charging_power", "re_ev_power_sum", "re_income_evs_cost", "re_income_evs_serve", "re_income_hys", "re_hy_cost",
         "re_hy_gen", "re_hydrogen_power_init", "re_hy_for_fc", "gen_hy", "re_income_evs_list_0", "re_income_evs_list_1",
         "re_income_evs_cost_list_0", "re_income_evs_cost_list_1"]


def attributes(env):
    out = []
    for name in ATTRS:
        if name[-2:] in ("_0", "_1") and name[:-2] in ("re_income_evs_list", "re_income_evs_cost_list"):
            out.append(float(getattr(env, name[:-2])[int(name[-1])]))
        else:
            v = getattr(env, name, np.nan)  # test_penalty does not exist before the first episode end (MGR:290)
            out.append(np.nan if v is None else float(v))
    return out


class Recorder:
    """Records every np.random.normal() draw and which OU channel consumed it."""

    def __init__(self):
        self.z = {}
        self.raw = []
        self._last = None
        self._orig = np.random.normal

        def normal(*a, **k):
            v = self._orig(*a, **k)
            self._last = float(v)
            self.raw.append(float(v))
            return v

        np.random.normal = normal

    def attach(self, env):
        for name, noise in (("pv", env.renew.pv_noise), ("wd", env.renew.wd_noise), ("price", env.price_noise)):
            self._wrap(name, noise)

    def _wrap(self, name, noise):
        orig = noise.sample

        def sample():
            out = orig()
            self.z[name] = self._last
            return out

        noise.sample = sample

    def take(self):
        z = [self.z.get("pv", np.nan), self.z.get("wd", np.nan), self.z.get("price", np.nan)]
        self.z = {}
        return z

    def close(self):
        np.random.normal = self._orig


def telemetry(env):
    h = env.hy_sys
    return [env.hy_act, h.hy_flow_speed, h.all_power_second, h.sty.Store_SOC, h.sty.capacity, h.hvs.total_mass_need,
            h.sty.hy_use, h.sty.not_meet, env.fc_power, env.hfc.hy_to_use, env.re_used_renew,
            env.re_ev_power_list[0], env.re_ev_power_list[1], env.re_hydrogen_power, env.income, env._last_reward,
            env.re_pv_power, env.re_wd_power, float(env.real_state[1]), h.hvs.arrive_number, h.hvs.line,
            len(h.hvs.needed_time_list)]


def station_block(env):
    out = []
    for st in env.env_aggregator.evcssp_evs_objects:
        sc = st._sc()
        out.append(sc[:6])
    return np.concatenate(out)


def run(name, kwargs, episodes, steps_per_episode, action_kind, seeds, py_seed=0, reseed_each_episode=False):
    random.seed(py_seed)
    np.random.seed(py_seed)
    rec = Recorder()
    ctor_seeds = None
    if seeds is not None:
        # the constructor consumes the two C++ streams as well (station constructors' evs_reset, the 101-step electrolyser
        # sweep with live FCEV demand, HYD:154-157): install known seeds in front of it so that hy_power_speed_list is
        # reproducible from the fixture alone
        ctor_seeds = (seeds[0] + 5000, seeds[1] + 5000)
        orclib.ref().ref_seed(*ctor_seeds)
    with contextlib.redirect_stdout(io.StringIO()):
        env = EvcsspManagerEnv_v6(**kwargs)
    # the constructor runs one reset() (MGR:120): its exogenous draws shape the OU states we start from
    ctor_days = [env.renew.pv_day, env.renew.wd_day]
    raw = list(rec.raw)
    from evcssp_env_cpp.envs.lion_cpp20.renewable import pv_power_data
    pv_drawn = pv_power_data[ctor_days[0]][0] > 0 and ctor_days[0] % 2 == 0
    ctor_z = [raw.pop(0) if pv_drawn else np.nan, raw.pop(0), raw.pop(0)]
    assert not raw
    rec.attach(env)
    rec.take()
    S = sum(kwargs["station_list"])
    rs = np.random.RandomState(1234 + S)
    D = int(env.observation_space.shape[0])
    hy_table = np.array(env.hy_sys.hy_power_speed_list, dtype=np.float64)
    data = dict(obs=[], reward=[], done=[], action=[], exo_z=[], telem=[], stations=[], reset_obs=[], reset_days=[],
                reset_z=[], reset_stations=[], slots0=[], slots1=[], seeds=[], reset_slots0=[], reset_slots1=[], attrs=[],
                real_state=[], reset_real_state=[], action_real=[], reset_attrs=[])
    hv_soc = []  # per step: the arrival SoCs hvs_step drew for its FCEV arrivals (HYD:258-260), in order
    for ep in range(episodes):
        if seeds is not None and (ep == 0 or reseed_each_episode):
            g, m = seeds[0] + ep, seeds[1] + ep
            orclib.ref().ref_seed(g, m)
            data["seeds"].append([ep, g, m])
        with contextlib.redirect_stdout(io.StringIO()):
            o = env.reset()
        data["reset_obs"].append(np.array(o, dtype=np.float64))
        data["reset_days"].append([env.renew.pv_day, env.renew.wd_day])
        data["reset_z"].append(rec.take())
        data["reset_stations"].append(station_block(env))
        data["reset_real_state"].append(np.array(env.real_state, dtype=np.float64))
        data["reset_attrs"].append([float(env.cumulated_income), float(env.cumulated_draw_ele), float(env.penalty)])
        data["reset_slots0"].append(env.env_aggregator.evcssp_evs_objects[0].slots())
        data["reset_slots1"].append(env.env_aggregator.evcssp_evs_objects[1].slots())
        for t in range(steps_per_episode):
            if action_kind == "none":
                act_in = None
                act = np.append(np.ones(S), [0, 0]).astype(np.float32)
            else:
                # dyadic grid (multiples of 2^-10) so f32 and f64 decodes of (a+1)/2 coincide
                act = (rs.randint(-1024, 1025, size=S + 2) / 1024.0).astype(np.float32)
                if action_kind == "random_hi":  # bias the electrolyser request high to hit the clamp
                    act[S] = np.float32(rs.randint(512, 1025) / 1024.0)
                act_in = act.copy()
            with contextlib.redirect_stdout(io.StringIO()):
                o, r, d, _ = env.step(act_in)
            env._last_reward = r
            data["action"].append(act)
            data["obs"].append(np.array(o, dtype=np.float64))
            data["reward"].append(r)
            data["done"].append(d)
            data["exo_z"].append(rec.take())
            data["telem"].append(telemetry(env))
            hv_soc.append([float(x) for x in env.hy_sys.hvs.arrive_soc_ini_list])
            assert len(hv_soc[-1]) == env.hy_sys.hvs.arrive_number
            data["stations"].append(station_block(env))
            data["attrs"].append(attributes(env))
            data["real_state"].append(np.array(env.real_state, dtype=np.float64))
            data["action_real"].append(np.array(env.action_real, dtype=np.float64))
            sts = env.env_aggregator.evcssp_evs_objects
            data["slots0"].append(sts[0].slots())
            data["slots1"].append(sts[1].slots())
    rec.close()
    out = {k: np.array(v) for k, v in data.items() if k not in ("slots0", "slots1", "reset_slots0", "reset_slots1")}
    for key in ("slots0", "slots1", "reset_slots0", "reset_slots1"):
        out[key] = np.array(data[key], dtype=np.float32)
    out["hv_soc"] = np.full((len(hv_soc), max(1, max(len(x) for x in hv_soc))), np.nan, dtype=np.float32)
    for i_, x in enumerate(hv_soc):
        out["hv_soc"][i_, :len(x)] = x
        assert np.array_equal(out["hv_soc"][i_, :len(x)].astype(np.float64), np.array(x)), "mk_soc returns C floats"
    out["hy_table"] = hy_table
    out["ctor_seeds"] = np.array(ctor_seeds if ctor_seeds is not None else (1, 1))  # c1: the process defaults (CHS:25,35-44)
    out["ctor_days"] = np.array(ctor_days)
    out["ctor_z"] = np.array(ctor_z)
    out["obs_dim"] = np.array(D)
    # what the constructor was actually given, and -- for kwargs left to their defaults -- the values the reference object ended up with
    # (HySystem: hydro_prod_rate None -> 430 HYD:140-143; HyStore: None -> 5000 HYD:96; HFC: None -> 100 HYD:401-404; MGR:25-27)
    out["ctor_kwargs_names"] = np.array(sorted(kwargs))
    eff = {"constant_charging": False, "seed_rand": True, "hydro_prod_rate": env.hy_sys.F_h_max, "hydro_store_vlt": env.hy_sys.sty.h_v_max,
           "init_soc": env.hy_sys.sty.init_soc_, "fc_max_power": env.hfc.cell_number, "fcev_permeate": env.fcev_permeate,
           "use_lagrange": False, "renew_fluctuate": env.renew.renew_fluctuate if hasattr(env.renew, "renew_fluctuate") else 0.0,
           "price_fluctuate": env.price_fluctuate, "hydro_loss": env.hy_sys.sty.hydro_loss}
    kw = dict(eff)
    kw.update(kwargs)
    for k_, v_ in eff.items():  # explicit kwargs must be what the object holds, too
        if k_ in kwargs and k_ not in ("seed_rand", "use_lagrange", "constant_charging"):
            assert float(kwargs[k_]) == float(v_), (k_, kwargs[k_], v_)
    out["kw_station_list"] = np.array(kw.pop("station_list"))
    out["kw_station_type"] = np.array([0 if t == "fast" else 1 for t in kw.pop("station_type_list")])
    for k, v in kw.items():
        out["kw_" + k] = np.array(float(v))
    out["episodes"] = np.array(episodes)
    out["steps_per_episode"] = np.array(steps_per_episode)
    out["telem_names"] = np.array(TELEM)
    out["attr_names"] = np.array(ATTRS)
    out["ret"] = np.array(float(np.sum(out["reward"][:steps_per_episode])))
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "obs_dim", D, "steps", len(out["reward"]), "return(ep0)", repr(float(out["ret"])),
          "final SOC", out["telem"][-1][3], "max queue", out["telem"][:, 21].max(), "min reset flow_in",
          min(out["reset_stations"][:, 5].min(), out["reset_stations"][:, 11].min()))
    return out


def base_kwargs(**over):
    kw = {"station_list": [20, 25], "station_type_list": ["fast", "slow"], "constant_charging": False,
          "seed_rand": False, "hydro_prod_rate": 100, "hydro_store_vlt": 500 / 20, "init_soc": 0.2,
          "fc_max_power": 100, "fcev_permeate": 0.01, "use_lagrange": False, "renew_fluctuate": 0.0,
          "price_fluctuate": 0.0, "hydro_loss": 0.0}
    kw.update(over)
    return kw


def main():
    # C1: the reference's own smoke test (test/env_test.py:14-49) -- default C++ seeds, action=None.
    # Must be first: it relies on the process-default stream state (srand never called, e seed 1).
    run("env_c1_envtest", base_kwargs(), 1, 96, "none", None, py_seed=0)
    # C3/C4 hub, random actions, explicit stream seeds, 2 episodes
    run("env_c3_random", base_kwargs(), 2, 96, "random", (101, 202), py_seed=1)
    # C2 hub: 16 fast, EV-only
    run("env_c2_random", base_kwargs(station_list=[16, 0], fcev_permeate=0.0), 2, 96, "random", (303, 404), py_seed=2)
    # C5 hub with fluctuations and loss
    run("env_c5_random", base_kwargs(station_list=[32, 32], renew_fluctuate=0.3, price_fluctuate=0.3,
                                     hydro_loss=0.02), 2, 96, "random", (505, 606), py_seed=3)
    # slow-only hub, high FCEV permeate (queue carries over), mid-episode reset (40-step episodes)
    run("env_slow_only_fcev", base_kwargs(station_list=[0, 25], fcev_permeate=0.05), 3, 40, "random", (707, 808),
        py_seed=4, reseed_each_episode=True)
    # big electrolyser on a big hub: exercises the grid-limit clamp branch (MGR:164-175)
    run("env_clamp", base_kwargs(station_list=[64, 64], hydro_prod_rate=2000, hydro_store_vlt=5000, init_soc=0.5),
        1, 96, "random_hi", (909, 1010), py_seed=5)
    # nearly full tank: upper_charge clamp (HYD:173-176), slow/fast order swapped
    run("env_full_tank", base_kwargs(station_list=[12, 10], station_type_list=["slow", "fast"], init_soc=0.95,
                                     hydro_store_vlt=5), 1, 96, "random", (1111, 1212), py_seed=6)
    # busy FCEV forecourt: several arrivals per step, the 15-minute FIFO carries cars over (HYD:266-279)
    run("env_fcev_queue", base_kwargs(fcev_permeate=0.06, hydro_store_vlt=200, init_soc=0.6), 1, 45, "random",
        (1515, 1616), py_seed=8)
    # constant-power fleet mode
    run("env_constant", base_kwargs(constant_charging=True), 1, 96, "random", (1313, 1414), py_seed=7)
    # a 3-pile fast station: init_station_car_number(mu = 1, 3) can come out negative and the fast station records it as
    # flow_in_number[-1] (CHS:1276, 832-842, 1617); many short episodes so that several resets do
    out = run("env_small_fast_neg", base_kwargs(station_list=[3, 0], fcev_permeate=0.0), 24, 6, "random", (1717, 1818),
              py_seed=9, reseed_each_episode=True)
    assert out["reset_stations"][:, 5].min() < 0, "no negative initial flow_in in this fixture: pick other seeds"
    # a forecourt whose 15-minute FIFO gets stuck: the waiting list grows to well over a hundred cars
    out = run("env_fcev_queue_deep", base_kwargs(fcev_permeate=0.08, hydro_store_vlt=400, init_soc=0.6), 1, 70, "random",
              (1919, 2020), py_seed=10)
    assert out["telem"][:, 21].max() > 64, out["telem"][:, 21].max()
    # stations of more than 64 piles (the reference takes any size, CHS:1148, 1458): 100 fast + 70 slow, a big electrolyser
    run("env_big_100_70", base_kwargs(station_list=[100, 70], hydro_prod_rate=2000, hydro_store_vlt=5000, init_soc=0.5,
                                      fcev_permeate=0.02), 2, 60, "random", (2121, 2222), py_seed=11)
    # ---- round 4, second batch (appended: the fixtures above come out as before) ----
    # both stations of the same kind (two slow stations; two fast ones with fluctuating exogenous series and tank loss)
    run("env_slow_slow", base_kwargs(station_list=[10, 12], station_type_list=["slow", "slow"]), 1, 96, "random", (2323, 2424), py_seed=12)
    run("env_fast_fast", base_kwargs(station_list=[8, 6], station_type_list=["fast", "fast"], renew_fluctuate=0.2, price_fluctuate=0.2,
                                     hydro_loss=0.01), 2, 60, "random", (2525, 2626), py_seed=13)
    # no electrolyser at all (Electrolyser.cell_number == 0, HYD:39-40): the tank only drains
    run("env_no_electrolyser", base_kwargs(hydro_prod_rate=0, init_soc=0.8, fcev_permeate=0.03), 1, 96, "random", (2727, 2828), py_seed=14)
    # a permeability above 1 falls back to 0.01 in the arrival lookup (CHS:765-775)
    run("env_permeate_cap", base_kwargs(fcev_permeate=1.5), 1, 48, "random", (2929, 3030), py_seed=15)
    # one pile per station
    run("env_one_pile", base_kwargs(station_list=[1, 1]), 2, 96, "random", (3131, 3232), py_seed=16)
    # constant-power fleet, slow / fast order swapped, fluctuating renewables
    run("env_constant_swapped", base_kwargs(station_list=[7, 9], station_type_list=["slow", "fast"], constant_charging=True,
                                            renew_fluctuate=0.1), 1, 96, "random", (3333, 3434), py_seed=17)
    # ---- round 5 (appended: the fixtures above come out as before) ----
    # stepping past `done` without a reset (MGR:271-299: the clock wraps, price_count keeps counting, the aggregator's price list keeps
    # growing AGG:147; evcssp_env_cpp/__init__.py:6 registers max_episode_steps=999): ONE episode of 250 steps on the C3 hub with
    # fluctuating series and tank loss, and one of 200 steps on the C2 hub
    run("env_past_done", base_kwargs(renew_fluctuate=0.3, price_fluctuate=0.3, hydro_loss=0.01), 1, 250, "random", (3535, 3636), py_seed=18)
    run("env_past_done_c2", base_kwargs(station_list=[16, 0], fcev_permeate=0.0), 1, 200, "random", (3737, 3838), py_seed=19)
    # every kwarg the reference gives a default left to it (MGR:25-27: a 430 m^3/h electrolyser, a 5000 m^3 tank at SOC 0.5, 100 fuel cells);
    # seed_rand=False only so that the streams are the recorded seeds'
    run("env_defaults", {"station_list": [20, 25], "station_type_list": ["fast", "slow"], "seed_rand": False}, 1, 96, "random", (3939, 4040), py_seed=20)
    # the tank at the two ends of the range the constructor accepts (HYD:137: 0.1 <= init_soc <= 1): at its 10 % floor from the first step on
    # (must_charge binds, the forecourt's demand is not met: HYD:172-176, 108-118) and full to the brim (upper_charge = 0: the electrolyser idles)
    out = run("env_tank_floor", base_kwargs(init_soc=0.1, fcev_permeate=0.03, hydro_store_vlt=5), 1, 96, "random", (4141, 4242), py_seed=21)
    assert out["telem"][:, 7].max() > 0, "no unmet demand in this fixture"
    run("env_tank_brim", base_kwargs(init_soc=1.0, hydro_store_vlt=5, station_list=[6, 9]), 1, 96, "random", (4343, 4444), py_seed=22)
    # ---- round 6 (appended: the fixtures above come out as before) ----
    # stations of more than 256 piles (the reference takes any count, CHS:1148, 1458): 300 fast + 270 slow -- a unit is walked in chunks of
    # 256 piles (k_slot_unit_any); evs_reset admits ~ S / 2 cars at once, the arrival SoCs' and stays' draws run across the chunks
    run("env_big_300_270", base_kwargs(station_list=[300, 270], hydro_prod_rate=2000, hydro_store_vlt=5000, init_soc=0.5,
                                       fcev_permeate=0.02), 2, 30, "random", (4545, 4646), py_seed=23)


if __name__ == "__main__":
    main()
