"""EvcsspManagerEnv_v6 -- single-environment drop-in for the reference class of the same name
(evcssp_env_cpp/envs/evcssp_manager.py:19-414) on top of the batched GPU runtime (N = 1).

Same constructor kwargs, ``reset() -> ndarray``, ``step(action) -> (ndarray, float, bool, {})``, ``seed``,
``action_space`` / ``observation_space`` bounds (MGR:74-118), ``render`` / ``close`` no-ops, and the telemetry
attributes trainers read after ``step`` (``re_*``, ``income``, ``fc_power``, ``hy_act`` ..., MGR:183-297).

Randomness: ``rng='compat'`` (default, like the reference) reproduces the reference's process-global streams
for this env -- glibc ``rand()`` + ``std::minstd_rand0`` on the device, while the exogenous draws stay on
the host exactly where the reference makes them: ``random.randint`` for the PV / wind day (REN:25,51-53) and
``np.random.normal()`` for the three OU processes (REN:73-74).  So ``random.seed(s); np.random.seed(s)``
before construction means what it means for the reference.  ``rng='philox'`` uses the device generator.
"""
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


class EvcsspManagerEnv_v6(object):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 30}

    def __init__(self, station_list, station_type_list, constant_charging=False, hydro_prod_rate=None,
                 hydro_store_vlt=None, seed_rand=True, init_soc=0.5, fc_max_power=None, fcev_permeate=0.01,
                 use_lagrange=False, renew_fluctuate=0, price_fluctuate=0, hydro_loss=0, rng="compat", device=0,
                 seed=None, data_dir=None):
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
                                   hydro_loss=hydro_loss)
        self._vec.set_telemetry(True)
        if rng == "compat":
            # Change_Use_Seed (MGR:28, CHS.hpp:27-41): seed_rand=True -> srand(time) on first use;
            # False -> rand() keeps glibc's default seed 1.  `e` always starts from its default seed 1 (CHS.hpp:25).
            if seed_rand:
                import time

                gseed = int(time.time()) & 0xFFFFFFFF
            else:
                gseed = 1
            self._vec.set_compat_seeds(np.array([[gseed, 1]], dtype=np.uint32))
            # the reference's constructor consumes stream draws before its reset() (station constructors,
            # HySystem's sweep): replay them so that an episode after construction matches the reference's
            self._vec.compat_replay_constructor()
        self.pile_number = [int(station_list[0]), int(station_list[1])]
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
        self._pv_day = 0
        self._wd_day = 0
        self._price_count = 0
        self._time = 0
        self.acumulate_reward = 0
        self.cumulated_income = 0
        self.cumulated_draw_ele = 0
        self.hy_init_soc = init_soc
        self.np_random = None
        self.seed()
        self.reset()  # MGR:120
        self.state = None

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed if seed is not None else 0)
        return [seed]

    # exogenous draws of one make_state (MGR:344-361), made on the host in the reference's order
    def _exo(self, time):
        z = np.zeros((1, 3))
        if self._rng != "compat":
            return None
        if self._pv[self._pv_day][time] > 0 and self._pv_day % 2 == 0:  # REN:40-41
            z[0, 0] = np.random.normal()
        z[0, 1] = np.random.normal()  # REN:47
        if self._price_count % 4 == 0:  # MGR:354
            z[0, 2] = np.random.normal()
        self._price_count += 1
        return z

    def reset(self):
        self.cumulated_income = 0
        self.cumulated_draw_ele = 0
        days = None
        if self._rng == "compat":
            self._pv_day = random.randint(0, 99)   # REN:51-53
            self._wd_day = random.randint(0, 149)
            days = np.array([[self._pv_day, self._wd_day]], dtype=np.int32)
        obs = self._vec.reset(days, self._exo(0))
        self._price_count = 0  # MGR:313
        self._time = 0
        self.state = self._vec.obs_f64()[0]
        return np.array(self.state)

    def step(self, action):
        S = sum(self.pile_number)
        if action is None:  # MGR:146-147
            action = np.append(np.ones(S), [0, 0], None)
        assert len(action) == S + 2  # MGR:148
        a = np.asarray(action, dtype=np.float32).reshape(1, S + 2)
        t_next = (self._time + 1) % 96
        self._vec.step(a, self._exo(t_next))
        self._time = t_next
        self.state = self._vec.obs_f64()[0]
        reward = float(self._vec.reward_f64()[0])
        done = bool(self._vec._done[0])
        tel = self._vec.telemetry()[0]
        for name, v in zip(_lib.TELEMETRY_NAMES, tel):
            setattr(self, "_t_" + name, float(v))
        # reference attribute names (MGR:175-297)
        self.hy_act = tel[0]
        self.re_hy_gen = 900.0 * tel[1]
        self.re_hydrogen_power_init = tel[2]
        self.fc_power = tel[8]
        self.re_hy_for_fc = tel[9]
        self.re_used_renew = tel[10]
        self.re_ev_power_list = [tel[11], tel[12]]
        self.re_hydrogen_power = tel[13]
        self.income = tel[14]
        self.re_pv_power = tel[16]
        self.re_wd_power = tel[17]
        self.cumulated_income += self.income
        self.acumulate_reward += reward
        self.deviation = abs(tel[3] - self.hy_init_soc)
        return self.state, reward, done, {}

    def render(self, mode='human'):
        pass

    def close(self):
        self._vec.close()

    def show_situation(self):
        return self._vec.slots()
