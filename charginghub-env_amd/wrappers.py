"""What RL trainers put around the hub: registration, time limit, vector-env adapters, telemetry in ``info``.

The reference registers one id with gym (``evcssp_env_cpp/__init__.py:3-8``: ``charging-hub-v6``,
``max_episode_steps=999``) and returns an empty ``info`` (``MGR:302``) while keeping the step's accounting on
attributes (``re_used_renew``, ``re_ev_power_list``, ``income`` ... ``MGR:183-297``).  Neither gym nor Gymnasium is a
dependency here: the adapters are duck-typed to the three calling conventions in use --

* ``make()``            -- ``gym.make('evcssp_env_cpp:charging-hub-v6', **kwargs)`` of the reference: the single drop-in
                           env inside a ``TimeLimit(999)``;
* ``HubVecEnv``         -- the stable-baselines ``VecEnv`` convention (``reset() -> obs``, ``step_async / step_wait``,
                           ``dones`` with automatic reset and ``terminal_observation`` in the per-env ``infos``);
* ``HubVectorEnv``      -- the Gymnasium ``VectorEnv`` convention (``reset(seed=, options=) -> (obs, info)``,
                           ``step -> (obs, reward, terminated, truncated, info)`` with dict-of-arrays ``info``).

All N envs of a ``VecChargingHub`` run in lock-step (one clock), so an episode end is an all-env event and automatic
reset is one ``chub_reset``.
"""
import numpy as np

from . import _lib
from .env import Box, EvcsspManagerEnv_v6, _space
from .vec_env import VecChargingHub

ENV_ID = "charging-hub-v6"
MAX_EPISODE_STEPS = 999   # evcssp_env_cpp/__init__.py:6
REWARD_THRESHOLD = 99     # evcssp_env_cpp/__init__.py:7

# reference attribute name (MGR:175-297) -> telemetry column of chub_get_telemetry
_T = {n: i for i, n in enumerate(_lib.TELEMETRY_NAMES)}
INFO_COLUMNS = {
    "hy_act": _T["hy_act"], "re_hydrogen_power_init": _T["all_power_second"], "Store_SOC": _T["Store_SOC"],
    "fc_power": _T["fc_power"], "re_hy_for_fc": _T["hy_to_use"], "re_used_renew": _T["re_used_renew"],
    "re_hydrogen_power": _T["re_hydrogen_power"], "income": _T["income"], "re_pv_power": _T["re_pv_power"],
    "re_wd_power": _T["re_wd_power"], "hy_use": _T["hy_use"], "not_meet": _T["not_meet"],
    "price_next": _T["price_next"], "fcev_arrive_number": _T["fcev_arrive_number"],
}


def telemetry_info(tel):
    """[N, T_COUNT] telemetry block -> dict of [N] arrays under the reference's attribute names"""
    info = {name: tel[:, col].copy() for name, col in INFO_COLUMNS.items()}
    info["re_hy_gen"] = 900.0 * tel[:, _T["hy_flow_speed"]]                                    # MGR:183
    info["re_ev_power_list"] = tel[:, [_T["re_ev_power_0"], _T["re_ev_power_1"]]].copy()       # MGR:212
    return info


class TimeLimit(object):
    """``gym.wrappers.TimeLimit`` as the reference's registration applies it (old 4-tuple step API): after
    ``max_episode_steps`` steps without a reset, ``done`` is forced and ``info['TimeLimit.truncated']`` says whether the
    env itself had not ended.  The hub's own ``done`` fires every 96 steps (MGR:271-273); it may be stepped on without a
    reset (infinite horizon), which is when this limit matters."""

    def __init__(self, env, max_episode_steps=MAX_EPISODE_STEPS):
        self.env = env
        self._max_episode_steps = int(max_episode_steps)
        self._elapsed_steps = None

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def step(self, action=None):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info = dict(info)
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, reward, done, info


def make(id=ENV_ID, **kwargs):
    """The reference's ``gym.make('evcssp_env_cpp:charging-hub-v6', **env_kwargs)`` (test/env_test.py:36)."""
    if id.split(":")[-1] != ENV_ID:
        raise ValueError("unknown environment id %r (have %r)" % (id, ENV_ID))
    return TimeLimit(EvcsspManagerEnv_v6(**kwargs), MAX_EPISODE_STEPS)


def register():
    """Register ``charging-hub-v6`` with whichever of gym / gymnasium is importable; returns the module names done."""
    done = []
    for modname in ("gym", "gymnasium"):
        try:
            mod = __import__(modname + ".envs.registration", fromlist=["register"])
        except Exception:
            continue
        mod.register(id=ENV_ID, entry_point="charginghub_env_amd:EvcsspManagerEnv_v6",
                     max_episode_steps=MAX_EPISODE_STEPS, reward_threshold=REWARD_THRESHOLD)
        done.append(modname)
    return done


def _spaces(vec, data_dir=None):
    price = np.fromfile((data_dir or _lib.DATA_DIR) + "/price_96.f64", dtype="<f8")
    obs_price = (price - np.mean(price)) / np.std(price)
    lo, hi = [-1.0, min(obs_price)], [1.0, max(obs_price)]           # MGR:51-104
    for k in range(2):
        if vec.piles[k] > 0:
            lo += [-1.0, -1.0, -1.0, 0]
            hi += [1.0, 1.0, 1.0, 2]
    lo += [0, 0, 0]
    hi += [1, 1, 1]
    return (_space(np.array(lo, dtype=np.float32), np.array(hi, dtype=np.float32)),
            _space(-1.0, 1.0, (vec.act_dim,)))


class _HubBatch(object):
    """shared by the two vector conventions: owns (or borrows) a VecChargingHub, counts steps, applies the limit"""

    def __init__(self, vec=None, n_envs=None, max_episode_steps=None, telemetry=False, data_dir=None, **hub_kwargs):
        if vec is None:
            if n_envs is None:
                raise ValueError("give either vec= or n_envs= and the hub kwargs")
            vec = VecChargingHub(n_envs, data_dir=data_dir, **hub_kwargs)
        self.vec = vec
        self.num_envs = vec.n_envs
        self.single_observation_space, self.single_action_space = _spaces(vec, data_dir)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space
        self.max_episode_steps = None if max_episode_steps is None else int(max_episode_steps)
        self.telemetry = bool(telemetry)
        if self.telemetry:
            vec.set_telemetry(True)
        self._elapsed = 0
        self.metadata = EvcsspManagerEnv_v6.metadata
        self.reward_range = (-float("inf"), float("inf"))
        self.spec = None

    def _step(self, actions):
        obs, reward, done, _ = self.vec.step(actions)
        self._elapsed += 1
        truncated = np.zeros(self.num_envs, dtype=bool)
        if self.max_episode_steps is not None and self._elapsed >= self.max_episode_steps:
            truncated = ~done
        tel = telemetry_info(self.vec.telemetry()) if self.telemetry else {}
        return obs, reward, done, truncated, tel

    def _reset(self):
        self._elapsed = 0
        return self.vec.reset()

    def close(self):
        self.vec.close()

    def render(self, mode="human"):
        return None


class HubVecEnv(_HubBatch):
    """stable-baselines ``VecEnv`` convention over one VecChargingHub.

    ``step_wait`` returns ``(obs, rewards, dones, infos)``; when the episode ends (all envs at once) the envs are reset,
    the returned ``obs`` is the first observation of the next episode and ``infos[i]['terminal_observation']`` holds the
    last one of the finished episode; ``infos[i]['TimeLimit.truncated']`` marks an end forced by ``max_episode_steps``.
    """

    def __init__(self, vec=None, n_envs=None, max_episode_steps=None, telemetry=False, **hub_kwargs):
        _HubBatch.__init__(self, vec, n_envs, max_episode_steps, telemetry, **hub_kwargs)
        self._pending = None

    def reset(self):
        return self._reset()

    def step_async(self, actions):
        self._pending = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, reward, done, truncated, tel = self._step(self._pending)
        self._pending = None
        dones = done | truncated
        infos = [dict() for _ in range(self.num_envs)]
        if tel:
            for name, col in tel.items():
                for i in range(self.num_envs):
                    infos[i][name] = col[i]
        if dones.any():
            # lock-step clock: the episode ends for every env in the same step
            for i in range(self.num_envs):
                infos[i]["terminal_observation"] = obs[i]
                infos[i]["TimeLimit.truncated"] = bool(truncated[i])
            obs = self._reset()
            dones = np.ones(self.num_envs, dtype=bool)
        return obs, reward, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        return [None] * self.num_envs  # streams are fixed by the hub's constructor seed (Philox key)

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self, attr_name)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [False] * n


class HubVectorEnv(_HubBatch):
    """Gymnasium ``VectorEnv`` convention over one VecChargingHub (autoreset mode "next step": the step after an
    episode end ignores its actions, resets and returns the first observation with zero reward)."""

    def __init__(self, vec=None, n_envs=None, max_episode_steps=None, telemetry=False, autoreset=True, **hub_kwargs):
        _HubBatch.__init__(self, vec, n_envs, max_episode_steps, telemetry, **hub_kwargs)
        self.autoreset = bool(autoreset)
        self._needs_reset = True
        self.closed = False

    def reset(self, seed=None, options=None):
        self._needs_reset = False
        return self._reset(), {}

    def step(self, actions):
        if self._needs_reset:
            if not self.autoreset:
                raise RuntimeError("episode has ended: call reset()")
            obs, info = self.reset()
            z = np.zeros(self.num_envs, dtype=bool)
            return obs, np.zeros(self.num_envs, dtype=np.float32), z, z.copy(), info
        obs, reward, done, truncated, info = self._step(actions)
        if (done | truncated).any():
            self._needs_reset = True
        return obs, reward, done, truncated, info

    def close(self, **kwargs):
        if not self.closed:
            self.closed = True
            _HubBatch.close(self)


class TorchHubVecEnv(object):
    """The hub for an on-device learner: actions in and observations / rewards / dones out are torch CUDA tensors, the
    step runs through the device-pointer entry points on torch's current stream, and nothing is copied through the host
    (the host-pointer ``VecChargingHub.step`` moves 47 + 15 floats per env over PCIe, about 3x the device time at 65 536
    envs).  ``step`` returns views of ONE packed [N, D + 2] output buffer (what ``chub_step_device_packed`` writes);
    with ``autoreset`` the envs are reset in the step that ends the episode and the returned observation is the first
    of the next one (``last_obs`` keeps the terminal one).  torch is imported here, not by the package."""

    def __init__(self, n_envs, station_list, station_type_list, seed=0, device=0, autoreset=True, **hub_kwargs):
        import torch  # before libchub is loaded by VecChargingHub: both must share one HIP runtime

        self.torch = torch
        self.device = torch.device("cuda", int(device))
        torch.cuda.set_device(self.device)
        self.vec = VecChargingHub(n_envs, station_list, station_type_list, seed=seed, rng="philox", device=int(device),
                                  **hub_kwargs)
        self.num_envs, self.obs_dim, self.act_dim = self.vec.n_envs, self.vec.obs_dim, self.vec.act_dim
        self.autoreset = bool(autoreset)
        self._packed = torch.empty((self.num_envs, self.obs_dim + 2), dtype=torch.float32, device=self.device)
        self._obs0 = torch.empty((self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device)
        self.last_obs = None
        self._t = 0

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def reset(self):
        self.vec.reset_device(self._obs0.data_ptr(), stream=self._stream())
        self._t = 0
        return self._obs0

    def step(self, actions):
        a = actions
        if not (a.is_cuda and a.dtype == self.torch.float32 and a.is_contiguous()
                and tuple(a.shape) == (self.num_envs, self.act_dim)):
            raise AssertionError("actions must be a contiguous float32 CUDA tensor of shape (%d, %d)"
                                 % (self.num_envs, self.act_dim))
        self.vec.step_device_packed(a.data_ptr(), self._packed.data_ptr(), stream=self._stream())
        return self._after_step()

    def _after_step(self):
        D = self.obs_dim
        obs, reward, done = self._packed[:, :D], self._packed[:, D], self._packed[:, D + 1] > 0.5
        self._t += 1
        if self.autoreset and self._t % 96 == 0:  # lock-step clock: done fires for every env in this step (MGR:271-273)
            self.last_obs = obs.clone()
            reward, done = reward.clone(), done.clone()
            obs = self.reset()
        return obs, reward, done, {}

    def pack_bits(self, actions):
        """[N, A] float32 CUDA action rows -> (pile_bits [N, W] int64, tail [N, 2] float32), on the device: what action_to_real
        (evcssp_manager.py:384-393) keeps of them (pile on iff a >= -2^-25).  A policy that samples on / off decisions can build
        the bits itself and never materialise the rows."""
        torch = self.torch
        S, W = self.vec.n_slots, self.vec.bit_words
        on = (actions[:, :S] >= -2.0 ** -25).to(torch.int64)
        if S < 64 * W:
            on = torch.nn.functional.pad(on, (0, 64 * W - S))
        sh = torch.arange(64, device=actions.device, dtype=torch.int64)
        bits = (on.view(-1, W, 64) << sh).sum(dim=2)  # bit 63 lands in the sign bit: the same 64 bits as the unsigned word
        return bits.contiguous(), actions[:, S:].contiguous()

    def step_bits(self, pile_bits, tail):
        """step() fed one bit per pile + the two tail floats (pack_bits' layout) instead of action rows: 8 W + 8 bytes of action input
        per env instead of 4 (S + 2); on the packed slot kernel the step reads the bits themselves"""
        torch, N, W = self.torch, self.num_envs, self.vec.bit_words
        if not (pile_bits.is_cuda and pile_bits.dtype == torch.int64 and pile_bits.is_contiguous() and tuple(pile_bits.shape) == (N, W)
                and tail.is_cuda and tail.dtype == torch.float32 and tail.is_contiguous() and tuple(tail.shape) == (N, 2)):
            raise AssertionError("pile_bits must be a contiguous int64 CUDA tensor of shape (%d, %d), tail float32 (%d, 2)" % (N, W, N))
        self.vec.step_bits_device_packed(pile_bits.data_ptr(), tail.data_ptr(), self._packed.data_ptr(), stream=self._stream())
        return self._after_step()

    def close(self):
        self.vec.close()


class StaggeredHub(object):
    """Non-lock-step episodes: G groups of envs whose days are offset against each other by 96/G slots.

    The N envs of a VecChargingHub that is only ever reset and stepped as a whole share one clock, so all N episodes end
    in the same step -- convenient for the kernels, but a learner then sees every env at the same time of day.  This front
    splits the env range into G contiguous groups (global env ids unchanged: group g covers ``[g*N/G, (g+1)*N/G)``) and keeps
    group g ``g * 96 // G`` slots ahead: ``reset()`` resets all groups and then walks group g through its head start with
    the all-on action (the reference's ``step(None)``, MGR:146-147).  ``step()`` steps every group, and a group whose day
    has ended (its own ``done``) is reset on the spot (``autoreset=True``): the returned observation rows of that group are
    the first of its next episode, ``info['terminal_observation']`` keeps the last ones and ``info['reset_groups']`` lists
    the groups reset in this step.

    Two forms.  The default keeps one hub per group: every group is bit-identical to a plain hub over the same global env
    range given the same head start.  ``one_handle=True`` runs all N envs in ONE hub with per-env clocks
    (``reset_envs`` / ``step_envs``, include/chub.h): one allocation, one launch per call whatever the number of groups
    (65 536 envs in 8 groups: 1.9 G env-steps/s on the device-pointer path); its Philox ticks count the handle's calls, so
    its random streams differ from the per-group form's (same distribution).
    """

    def __init__(self, n_envs, groups, station_list, station_type_list, seed=0, env_id0=0, autoreset=True,
                 hub_factory=None, one_handle=False, **hub_kwargs):
        groups = int(groups)
        if groups < 1 or n_envs % groups:
            raise ValueError("n_envs must split evenly over the groups")
        if groups > 96:
            raise ValueError("at most 96 groups (one per slot of the day)")
        make = hub_factory or VecChargingHub
        self.n_envs, self.groups, self.per = int(n_envs), groups, int(n_envs) // groups
        self.offsets = [g * 96 // groups for g in range(groups)]
        self.hub = None
        if one_handle:
            self.hub = make(self.n_envs, station_list, station_type_list, seed=seed, env_id0=env_id0, **hub_kwargs)
            self.hubs = [self.hub]
        else:
            self.hubs = [make(self.per, station_list, station_type_list, seed=seed, env_id0=env_id0 + g * self.per,
                              **hub_kwargs) for g in range(groups)]
        self.obs_dim, self.act_dim = self.hubs[0].obs_dim, self.hubs[0].act_dim
        self.piles, self.n_slots = self.hubs[0].piles, self.act_dim - 2
        self.autoreset = bool(autoreset)

    def _rows(self, g):
        return slice(g * self.per, (g + 1) * self.per)

    def _mask(self, gs):
        m = np.zeros(self.n_envs, dtype=bool)
        for g in gs:
            m[self._rows(g)] = True
        return m

    @property
    def clocks(self):
        """slot of day of every group"""
        if self.hub is not None:
            return [int(t) for t in self.hub.env_clocks()[::self.per]]
        return [h.clock for h in self.hubs]

    def reset(self):
        if self.hub is not None:
            head = np.zeros((self.n_envs, self.act_dim), dtype=np.float32)
            head[:, :self.n_slots] = 1.0  # step(None): every pile on, fuel cell and electrolyser at 0 (MGR:146-147, 384-404)
            obs = self.hub.reset()
            for k in range(1, max(self.offsets) + 1):  # the k-th head-start step: every group that is at least k slots ahead
                obs = self.hub.step_envs(self._mask([g for g in range(self.groups) if self.offsets[g] >= k]), head)[0]
            return obs
        obs = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        head = np.zeros((self.per, self.act_dim), dtype=np.float32)
        head[:, :self.n_slots] = 1.0  # step(None): every pile on, fuel cell and electrolyser at 0 (MGR:146-147, 384-404)
        for g, h in enumerate(self.hubs):
            o = h.reset()
            for _ in range(self.offsets[g]):
                o = h.step(head)[0]
            obs[self._rows(g)] = o
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.n_envs, self.act_dim):  # MGR:148
            raise AssertionError("actions must have shape (%d, %d)" % (self.n_envs, self.act_dim))
        info = {}
        if self.hub is not None:
            obs, reward, done, _ = self.hub.step(a)
            ended = [g for g in range(self.groups) if done[self._rows(g)].all()]
            if ended and self.autoreset:
                info["terminal_observation"] = np.zeros_like(obs)
                m = self._mask(ended)
                info["terminal_observation"][m] = obs[m]
                info["reset_groups"] = ended
                obs = self.hub.reset_envs(m)
            return obs, reward, done, info
        obs = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        reward = np.zeros(self.n_envs, dtype=np.float32)
        done = np.zeros(self.n_envs, dtype=bool)
        for g, h in enumerate(self.hubs):
            rows = self._rows(g)
            o, r, d, _ = h.step(a[rows])
            reward[rows], done[rows] = r, d
            if d.all() and self.autoreset:
                info.setdefault("terminal_observation", np.zeros_like(obs))[rows] = o
                info.setdefault("reset_groups", []).append(g)
                o = h.reset()
            obs[rows] = o
        return obs, reward, done, info

    def set_telemetry(self, on=True):
        for h in self.hubs:
            h.set_telemetry(on)

    def telemetry(self):
        return np.concatenate([h.telemetry() for h in self.hubs], axis=0)

    def close(self):
        for h in self.hubs:
            h.close()


__all__ = ["ENV_ID", "StaggeredHub", "TorchHubVecEnv", "MAX_EPISODE_STEPS", "TimeLimit", "make", "register", "HubVecEnv", "HubVectorEnv",
           "telemetry_info", "Box"]
