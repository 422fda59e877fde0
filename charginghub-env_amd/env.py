"""EvcsspManagerEnv_v6 -- single-environment drop-in for the reference class of the same name
(evcssp_env_cpp/envs/evcssp_manager.py:19-414) on top of the batched GPU runtime (N = 1).

Same constructor kwargs, ``reset() -> ndarray``, ``step(action) -> (ndarray, float, bool, {})``, ``seed``,
``action_space`` / ``observation_space`` bounds (MGR:74-118), ``render`` / ``close`` no-ops, and every attribute the reference
sets in ``reset`` / ``step`` that a trainer or an evaluation script reads afterwards -- and read-only views of the sub-objects such
scripts reach into (``env.hy_sys.sty.Store_SOC``, ``env.hy_sys.hvs.total_mass_need``, ``env.hfc.hy_to_use``,
``env.env_aggregator.evcssp_evs_objects[k].charge_power``, ``env.renew.pv_day`` ...): ``real_state`` (MGR:372), ``action_real``,
``re_*``, ``income``, ``fc_power``, ``hy_act``, ``gen_hy`` (MGR:150-231), ``cumulated_income`` / ``cumulated_draw_ele`` (MGR:259-262),
``acumulate_reward``, ``deviation``, ``test_penalty`` at the end of an episode (MGR:275-297), ``penalty`` / ``lagrangian_factor``
(MGR:128, 314-315).  tests/test_gpu_parity.py holds them to the values recorded from the reference.

One ``step()`` is ONE call into libchub and no device read: the actions go up from a pinned buffer, the kernels write the observation,
reward, done flag and the telemetry block straight into pinned host memory (chub_telemetry_host), and everything above is read
from there.

Randomness: ``rng='compat'`` (default, like the reference) reproduces the reference's process-global streams
for this env -- glibc ``rand()`` + ``std::minstd_rand0`` on the device, while the exogenous draws stay on
the host exactly where the reference makes them: ``random.randint`` for the PV / wind day (REN:25,51-53) and
``np.random.normal()`` for the three OU processes (REN:73-74).  So ``random.seed(s); np.random.seed(s)``
before construction means what it means for the reference.  ``rng='philox'`` uses the device generator.
"""
import ctypes as C
import os
import random

import numpy as np

from . import _lib
from .vec_env import VecChargingHub


class Box(object):
    """Minimal stand-in for gym.spaces.Box (low / high / shape / dtype / sample / contains)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))


def _space(low, high, shape=None):
    try:
        from gym import spaces

        return spaces.Box(low=low, high=high, shape=shape, dtype=np.float32)
    except Exception:
        return Box(low, high, shape, np.float32)


class _QueueLen(object):
    """stands in for HyFCEVStation.needed_time_list / needed_hy_list (HYD:264-265): the reference's scripts only ever take its len()"""

    def __init__(self, n):
        self._n = int(n)

    def __len__(self):
        return self._n


class _StationView(object):
    """read-only stand-in for env.env_aggregator.evcssp_evs_objects[k] (the Boost-bound Fast / SlowChargeStation, MAIN:186-290): the
    scalars the reference's host reads off a station (AGG:198-218, MGR:364-368)"""

    def __init__(self, env, k):
        self._env, self._k = env, k
        self.charge_number = env.pile_number[k]
        const_power = 36.44764034125146 if env._types[k] == "fast" else 5.254973139368931
        self.transformer_limit = float(np.float32(np.float32(const_power) * np.float32(self.charge_number)))  # CHS.hpp:1133-1134 / 1443-1444

    def _col(self, name):
        return self._env._last_tel[_lib.T["%s_%d" % (name, self._k)]]

    min_power = property(lambda self: self._col("min_power"))
    charge_power = property(lambda self: self._col("charge_power"))
    max_power = property(lambda self: self._col("max_power"))
    line = property(lambda self: int(self._col("line")))
    flow_in_number = property(lambda self: [int(self._col("flow_in"))])         # the reference reads [-1] (AGG:218)
    station_time_hole = property(lambda self: self._env._time)
    car_number = property(lambda self: int(self._env._vec.station_scalars()[0, self._k, 3]))  # (one device read: not in the telemetry block)


class _Views(object):
    """read-only stand-ins for the sub-objects evaluation scripts reach into: env.env_aggregator (AGG), env.hy_sys / .sty / .hvs /
    .ele (HYD), env.hfc (HYD:394-430), env.renew (REN) -- attribute names of the reference, values from the telemetry block of the
    last reset() / step()"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _live(cls_name, fields):
    """a small class whose attributes are computed from the env's last telemetry row on access"""
    return type(cls_name, (object,), dict({"__init__": lambda self, env: setattr(self, "_env", env)},
                                          **{k: property(f) for k, f in fields.items()}))


_T = _lib.T
_Store = _live("HyStoreView", {
    "Store_SOC": lambda s: s._env._last_tel[_T["Store_SOC"]], "capacity": lambda s: s._env._last_tel[_T["capacity"]],
    "hy_use": lambda s: s._env._last_tel[_T["hy_use"]], "not_meet": lambda s: s._env._last_tel[_T["not_meet"]],
    "capacity_mass": lambda s: s._env._capacity_mass, "init_soc_": lambda s: s._env.hy_init_soc,
    "h_v_max": lambda s: s._env._capacity_mass / (0.089 * (200 / 1)) / 1000, "density": lambda s: 0.089 * (200 / 1)})
_Hvs = _live("HyFCEVStationView", {
    "total_mass_need": lambda s: s._env._last_tel[_T["total_mass_need"]], "arrive_number": lambda s: int(s._env._last_tel[_T["fcev_arrive_number"]]),
    "line": lambda s: int(s._env._last_tel[_T["fcev_line"]]), "needed_time_list": lambda s: _QueueLen(s._env._last_tel[_T["fcev_queue_len"]]),
    "needed_hy_list": lambda s: _QueueLen(s._env._last_tel[_T["fcev_queue_len"]])})
_Hfc = _live("HFCView", {"hy_to_use": lambda s: s._env._last_tel[_T["hy_to_use"]], "cell_number": lambda s: s._env._fc_cells})
_Renew = _live("ReNewView", {"pv_day": lambda s: int(s._env._last_tel[_T["pv_day"]]), "wd_day": lambda s: int(s._env._last_tel[_T["wd_day"]])})


class _HySys(object):
    def __init__(self, env):
        self._env = env
        self.sty, self.hvs = _Store(env), _Hvs(env)

    hy_flow_speed = property(lambda s: s._env._last_tel[_T["hy_flow_speed"]])
    hy_flow_speed_15 = property(lambda s: 15 * 60 * s._env._last_tel[_T["hy_flow_speed"]])
    all_power_second = property(lambda s: s._env._last_tel[_T["all_power_second"]])
    all_power_15 = property(lambda s: s._env._last_tel[_T["all_power_second"]] * 15 * 60)
    sys_time = property(lambda s: s._env._time)
    hy_power_speed_list = property(lambda s: s._env._vec.hy_table(env=0).tolist())               # HYD:154-157
    hy_power_speed_list_input = property(lambda s: [0.01 * i for i in range(101)] + [0.01 * 100])


class _Aggregator(object):
    def __init__(self, env, station_list, station_type_list, constant_charging):
        self._env = env
        self.station_list = self.pile_number = list(station_list)
        self.station_type_list = list(station_type_list)
        self.station_number = 2
        self.constant_charging, self.wait = constant_charging, True
        self.price_constant = list(env._price)
        self.price = [] + self.price_constant
        self.price_max, self.price_min = max(self.price), min(self.price)
        self.evcssp_evs_objects = [_StationView(env, 0), _StationView(env, 1)]
        self.total_max_power = 0
        for station in self.evcssp_evs_objects:
            self.total_max_power += station.transformer_limit

    aggregator_time_hole = property(lambda s: s._env._time)
    evcssp_charge_power = property(lambda s: [st.charge_power for st in s.evcssp_evs_objects])
    evcssp_max_demand = property(lambda s: [st.max_power for st in s.evcssp_evs_objects])
    evcssp_min_demand = property(lambda s: [st.min_power for st in s.evcssp_evs_objects])
    ag_flow_in_number = property(lambda s: [st.flow_in_number[-1] for st in s.evcssp_evs_objects])
    ag_car_number = property(lambda s: [st.car_number for st in s.evcssp_evs_objects])
    ag_full = property(lambda s: [st.car_number >= st.charge_number for st in s.evcssp_evs_objects])


class EvcsspManagerEnv_v6(object):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 30}

    def __init__(self, station_list, station_type_list, constant_charging=False, hydro_prod_rate=None,
                 hydro_store_vlt=None, seed_rand=True, init_soc=0.5, fc_max_power=None, fcev_permeate=0.01,
                 use_lagrange=False, renew_fluctuate=0, price_fluctuate=0, hydro_loss=0, rng="compat", device=0,
                 seed=None, data_dir=None, compat_seeds=None):
        """Beyond the reference's kwargs (MGR:25-27): rng / device / seed / data_dir, and compat_seeds = (srand seed, e.seed()):
        the state of the reference's two process-global C++ streams in front of the constructor (default: what
        Change_Use_Seed(seed_rand) leaves, CHS.hpp:25-41)."""
        assert len(station_list) == len(station_type_list) == 2  # MGR:37
        self.fcev_permeate = fcev_permeate
        self.init_soc = init_soc
        self.price_fluctuate = price_fluctuate
        self.renew_fluctuate = renew_fluctuate
        self.simulate = False
        self._rng = rng
        self._vec = VecChargingHub(1, station_list, station_type_list, seed=0 if seed is None else seed, rng=rng,
                                   device=device, data_dir=data_dir, constant_charging=constant_charging,
                                   hydro_prod_rate=hydro_prod_rate, hydro_store_vlt=hydro_store_vlt,
                                   init_soc=init_soc, fc_max_power=fc_max_power, fcev_permeate=fcev_permeate,
                                   renew_fluctuate=renew_fluctuate, price_fluctuate=price_fluctuate,
                                   hydro_loss=hydro_loss, copy_outputs=False)
        self._vec.set_telemetry(True)
        if rng == "compat":
            # Change_Use_Seed (MGR:28, CHS.hpp:27-41): seed_rand=True -> srand(time) on first use;
            # False -> rand() keeps glibc's default seed 1.  `e` always starts from its default seed 1 (CHS.hpp:25).
            if compat_seeds is not None:
                gseed, eseed = int(compat_seeds[0]), int(compat_seeds[1])
            elif seed_rand:
                import time

                gseed, eseed = int(time.time()) & 0xFFFFFFFF, 1
            else:
                gseed, eseed = 1, 1
            self._vec.set_compat_seeds(np.array([[gseed, eseed]], dtype=np.uint32))
            # the reference's constructor consumes stream draws before its reset() (station constructors,
            # HySystem's sweep): replay them so that an episode after construction matches the reference's
            self._vec.compat_replay_constructor()
        self.pile_number = [int(station_list[0]), int(station_list[1])]
        self._types = list(station_type_list)
        self._fc_cells = 100 if fc_max_power is None else fc_max_power  # HFC.cell_number (HYD:401-404)
        data = data_dir or _lib.DATA_DIR
        self._price = np.fromfile(data + "/price_96.f64", dtype="<f8")
        self._pv = np.fromfile(data + "/pv_100x96.f64", dtype="<f8").reshape(100, 96)
        self._wd = np.fromfile(data + "/wd_150x96.f64", dtype="<f8").reshape(150, 96)
        self.price_mean = np.mean(self._price)
        self.price_std = np.std(self._price)
        obs_price = (self._price - self.price_mean) / self.price_std
        # observation / action bounds, MGR:51-118
        min_list = [-1.0, min(obs_price)]
        max_list = [1.0, max(obs_price)]
        n_active = 1 if (self.pile_number[0] == 0 or self.pile_number[1] == 0) else 2
        for _ in range(n_active):
            min_list += [-1.0, -1.0, -1.0, 0]
            max_list += [1.0, 1.0, 1.0, 2]
        min_list += [0, 0, 0]
        max_list += [1, 1, 1]
        self.low_state = np.array(min_list, dtype=np.float32)
        self.high_state = np.array(max_list, dtype=np.float32)
        self.viewer = None
        self.action_space = _space(-1.0, 1.0, (sum(self.pile_number) + 2,))
        self.observation_space = _space(self.low_state, self.high_state)
        if rng == "compat":
            # side effect of the reference's constructor: each of its three OU_Noise objects calls
            # random.seed(1) (REN:63 via MGR:30 and REN:34-35), so the first randint() pair of reset()
            # always starts from that state
            random.seed(1)
        # ---- what one step() touches, set up once: the pinned action row, the exogenous draws, raw pointers, live views of the
        # handle's pinned outputs and telemetry block
        v = self._vec
        self._S = sum(self.pile_number)
        self._act = v.pinned_actions()                   # [1, S + 2] f32, pinned
        self._z = v._pinned_array((1, 3), np.float64)    # the step's exogenous normals (COMPAT), pinned: read by the kernel where they are
        tel, obs64, rew64 = v.telemetry_views()
        self._tel, self._obs64, self._rew64 = tel[:, 0], obs64[0], rew64  # this env's column / row of the block: live views
        self._p = [C.c_void_p(a.ctypes.data) for a in (self._act, self._z, v._obs, v._reward, v._done)]
        self._chub_step = v._lib.chub_step
        self._h = v._h
        T = _lib.T
        # real_state (MGR:364-370) as telemetry columns: [time, price_next, {min, charge, max power, line} per station with piles,
        # Store_SOC, pv power, wind power]; entry 0 (a placeholder column here) is overwritten with the slot of day
        cols = [0, T["price_next"]]
        for k in (0, 1):
            if self.pile_number[k] > 0:
                cols += [T["min_power_%d" % k], T["charge_power_%d" % k], T["max_power_%d" % k], T["line_%d" % k]]
        self._rs_cols = np.array(cols + [T["Store_SOC"], T["re_pv_power"], T["re_wd_power"]])
        # HyStore (HYD:92-100): capacity_mass in g
        self._capacity_mass = (0.089 * (200 / 1)) * ((5000 if hydro_store_vlt is None else hydro_store_vlt) * 1000)
        self.real_time_max = 96
        self.scale_pv = 5
        self.scale_wd = 1
        self._pv_day = 0
        self._wd_day = 0
        self._price_count = 0
        self._time = 0
        self.acumulate_reward = 0
        self.cumulated_income = 0
        self.cumulated_draw_ele = 0
        self.hy_init_soc = init_soc
        self._last_tel = [0.0] * _lib.T_COUNT
        # the sub-objects of the reference class that scripts reach into, as read-only views (values of the last reset() / step())
        self.env_aggregator = _Aggregator(self, station_list, station_type_list, constant_charging)
        self.hy_sys, self.hfc, self.renew = _HySys(self), _Hfc(self), _Renew(self)
        self.np_random = None
        self.seed()
        self.reset()  # MGR:120
        self.real_state = []  # MGR:121-123: the constructor leaves these empty until the caller's own reset()
        self.np_random = None
        self.state = None
        self.acumulate_reward = 0
        self.use_lagrangian = False
        self.lagrangian_factor = None
        self.penalty = 0
        self.cumulated_income = 0
        self.cumulated_draw_ele = 0

    def seed(self, seed=None):
        # MGR:132-134: `self.np_random, seed = seeding.np_random(seed); return [seed]` -- gym generates a seed when none is given and
        # returns the one it used (the simulation itself never draws from np_random)
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "big")
        self.np_random = np.random.RandomState(int(seed) % (1 << 32))
        return [seed]

    def set_compat_seeds(self, glibc_seed, minstd_seed):
        """rng='compat': install (srand seed, e.seed()) for this env's two reference streams -- what re-seeding the reference's
        process-global generators in front of a reset() does"""
        self._vec.set_compat_seeds(np.array([[int(glibc_seed), int(minstd_seed)]], dtype=np.uint32))

    # exogenous draws of one make_state (MGR:344-361), made on the host in the reference's order
    def _exo(self, time):
        if self._rng != "compat":
            return None
        z = self._z
        z[0, 0] = z[0, 1] = z[0, 2] = 0.0
        if self._pv[self._pv_day][time] > 0 and self._pv_day % 2 == 0:  # REN:40-41
            z[0, 0] = np.random.normal()
        z[0, 1] = np.random.normal()  # REN:47
        if self._price_count % 4 == 0:  # MGR:354
            z[0, 2] = np.random.normal()
        self._price_count += 1
        return z

    def _make_state(self, time, tel=None):
        """what make_state (MGR:344-373) leaves: real_state, state, the renewable powers and the price noise"""
        self._last_tel = tel = self._tel.tolist() if tel is None else tel
        rs = self._tel[self._rs_cols]  # fancy indexing: a fresh array
        rs[0] = time
        self.real_state = rs
        self.re_pv_power = tel[16]
        self.re_wd_power = tel[17]
        self.state = np.array(self._obs64)

    def _live(self):
        if self.__dict__.get("_h") is None:
            raise _lib.ChubError("this EvcsspManagerEnv_v6 has been closed (its libchub handle and pinned buffers are gone)")

    def reset(self):
        self._live()
        self.cumulated_income = 0
        self.cumulated_draw_ele = 0
        self.real_state = []
        self.accumulate_reward = 0  # (sic, MGR:308: acumulate_reward is never reset)
        days = None
        if self._rng == "compat":
            self._pv_day = random.randint(0, 99)   # REN:51-53
            self._wd_day = random.randint(0, 149)
            days = np.array([[self._pv_day, self._wd_day]], dtype=np.int32)
        z = self._exo(0)
        self._vec.reset(days, None if z is None else z.copy())
        self._price_count = 0  # MGR:313
        self._time = 0
        self.env_aggregator.price = [] + self.env_aggregator.price_constant  # AGG:171
        self._make_state(0)
        self.lagrangian_factor = None
        self.penalty = 0
        return np.array(self.state)

    def step(self, action):
        self._live()
        S = self._S
        if action is None:  # MGR:146-147
            action = np.append(np.ones(S), [0, 0], None)
        assert len(action) == S + 2  # MGR:148
        self.action_real = self.action_to_real(action)  # MGR:149-151
        self._act[0, :] = action
        t_next = (self._time + 1) % 96
        p = self._p
        rc = self._chub_step(self._h, p[0], p[1] if self._exo(t_next) is not None else None, p[2], p[3], p[4])
        if rc:
            _lib.check(rc)
        # chub_step returns completed: observation, reward, done and the telemetry block are in pinned host memory now
        tel = self._tel.tolist()  # one conversion of the 38 columns to Python floats
        reward = float(self._rew64[0])
        done = bool(self._vec._done[0])
        # reference attribute names (MGR:175-297)
        self.hy_act = tel[0]
        self.gen_hy = tel[1] > 0.5                            # MGR:161,173-179
        self.re_hy_gen = 15 * 60 * tel[1]                     # hy_flow_speed_15, HYD:183
        self.re_hydrogen_power_init = tel[2]
        self.re_ev_power_list = [tel[11], tel[12]]
        self.re_used_renew = tel[10]
        self.fc_power = tel[8]
        self.re_hy_for_fc = tel[9]
        ev0, ev1, hydrogen_power = tel[24], tel[25], tel[13]
        self.real_charging_power = self.re_ev_power_sum = tel[26]
        self.re_hydrogen_power = hydrogen_power
        real_price_dollar = tel[27] / 4                       # MGR:234
        self.re_price_dollar = real_price_dollar
        P0, P1 = tel[29], tel[34]
        self.re_income_evs_cost = - real_price_dollar * (ev0 + ev1)
        income_evs_serve = 0.8 * (int(tel[32]) + int(tel[37]))
        self.re_income_evs_cost_list = [- real_price_dollar * ev0, - real_price_dollar * ev1]
        self.re_income_evs_list = [0.42 / 4 * P0, 0.21 / 4 * P1]
        self.re_income_evs_serve = income_evs_serve
        self.re_income_hys = 6 / 1000 * tel[6]
        self.re_hy_cost = -real_price_dollar * hydrogen_power
        self.income = tel[14]
        self.cumulated_income += self.income
        self.cumulated_draw_ele += ev0 + ev1 + hydrogen_power  # MGR:262
        self.acumulate_reward += reward
        store_soc = tel[3]
        if done:  # MGR:275-297 (its prints left out)
            temp_deviation = abs(store_soc - self.hy_init_soc) * self._capacity_mass / 1000
            temp_deviation = temp_deviation / 0.2
            self.test_penalty = abs(temp_deviation)
        self.deviation = abs(store_soc - self.hy_init_soc)
        self.env_aggregator.price.append(self.env_aggregator.price_constant[self._time])  # AGG:147
        self._time = t_next
        self._make_state(t_next, tel)
        return self.state, reward, done, {}

    def __getattr__(self, name):
        # every telemetry column of the last step by name (env._t_Store_SOC, env._t_hy_use ...: _lib.TELEMETRY_NAMES), read from the
        # pinned block on demand, for scripts that want more than the reference exposes
        if name.startswith("_t_") and name[3:] in _lib.T:
            tel = self.__dict__.get("_tel")  # (absent while the constructor is still running, gone after close())
            if tel is None:
                raise AttributeError("%s: no telemetry block (the env is not constructed yet, or closed)" % name)
            return float(tel[_lib.T[name[3:]]])
        raise AttributeError(name)

    @staticmethod
    def action_to_real(action):
        """MGR:384-404: one bit per pile, the two tail actions swapped (the same values as the reference's element-by-element loop,
        computed on the whole row at once)"""
        if action[-1] is None or action[-2] is None:  # the reference passes a None through (an object array)
            real_action = (np.array(action[:-2]) + 1) / 2
            real_actions = [1.0 if act >= 0.5 else 0.0 for act in real_action]
            real_actions.append((action[-1] + 1) / 2 if action[-1] is not None else action[-1])
            real_actions.append((action[-2] + 1) / 2 if action[-2] is not None else action[-2])
            return np.array(real_actions)
        n = len(action) - 2
        out = np.empty(n + 2)
        out[:n] = (np.asarray(action[:-2]) + 1) / 2 >= 0.5
        out[n] = (action[-1] + 1) / 2
        out[n + 1] = (action[-2] + 1) / 2
        return out

    def render(self, mode='human'):
        pass

    def close(self):
        # the views and raw pointers into handle-owned pinned memory go first: after chub_destroy they would dangle
        for name in ("_tel", "_obs64", "_rew64", "_p", "_h", "_act", "_z"):
            self.__dict__[name] = None
        vec = self.__dict__.get("_vec")
        if vec is not None:
            vec.close()

    def show_situation(self):
        return self._vec.slots()
