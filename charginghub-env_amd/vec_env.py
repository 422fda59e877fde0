"""VecChargingHub -- N lock-step charging-hub environments on one MI355X.

Batched counterpart of ``EvcsspManagerEnv_v6`` (evcssp_manager.py:19-414): same constructor kwargs, same
observation / action layout per env, arrays of shape ``[N, ...]``.  All simulation runs in libchub's HIP
kernels; this class only marshals numpy arrays across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ChubConfig, ChubError, ChubOptions, check, load_library


def make_config(station_list, station_type_list, constant_charging=False, hydro_prod_rate=None, hydro_store_vlt=None,
                init_soc=0.5, fc_max_power=None, fcev_permeate=0.01, renew_fluctuate=0, price_fluctuate=0,
                hydro_loss=0):
    """Constructor kwargs of the reference (MGR:25-27) -> chub_config, with the reference's defaults
    (hydro_prod_rate None -> 430 HYD:140-143, hydro_store_vlt None -> 5000 HYD:96, fc_max_power None -> 100 HYD:401-404)."""
    if not (len(station_list) == len(station_type_list) == 2):  # MGR:37
        raise AssertionError("station_list and station_type_list must both have 2 entries")
    cfg = ChubConfig()
    for k in range(2):
        cfg.station_list[k] = int(station_list[k])
        if station_type_list[k] == "fast":
            cfg.station_type_list[k] = _lib.CHUB_FAST
        elif station_type_list[k] == "slow":
            cfg.station_type_list[k] = _lib.CHUB_SLOW
        else:
            raise ValueError("EVS type must be fast or slow")  # AGG:196
    cfg.constant_charging = int(bool(constant_charging))
    cfg.hydro_prod_rate = 430.0 if hydro_prod_rate is None else float(hydro_prod_rate)
    cfg.hydro_store_vlt = 5000.0 if hydro_store_vlt is None else float(hydro_store_vlt)
    cfg.init_soc = float(init_soc)
    cfg.fc_max_power = 100.0 if fc_max_power is None else float(fc_max_power)
    cfg.fcev_permeate = float(fcev_permeate)
    cfg.renew_fluctuate = float(renew_fluctuate)
    cfg.price_fluctuate = float(price_fluctuate)
    cfg.hydro_loss = float(hydro_loss)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class VecChargingHub(object):
    def __init__(self, n_envs, station_list, station_type_list, seed=0, rng="philox", device=0, env_id0=0,
                 data_dir=None, slot_kernel="auto", no_arena=False, copy_outputs=True, fused_step="auto", tile="auto", walk_ahead="auto",
                 work_order="auto", span_steps="auto", span_tails="auto", **kwargs):
        """slot_kernel: PHILOX handles: "auto" (the packed slot kernel wherever the hub shape allows), "wave" (the wave-local one for
        every step) or "packed"; COMPAT handles: "auto" / "packed" the split step (stream walks one env per lane, one slot pass over both
        stations), "wave" one kernel per station (bit-identical); no_arena: one device allocation per array (no snapshots) -- chub_options.
        copy_outputs=False: reset() / step() return the handle's own pinned arrays (valid until the next call) instead of
        fresh copies -- at 65 536 envs the copies are a third of a host-pointer step."""
        kwargs.pop("seed_rand", None)
        kwargs.pop("use_lagrange", None)  # ignored by the reference too (MGR:126)
        self._lib = load_library()
        self.cfg = make_config(station_list, station_type_list, **kwargs)
        self.n_envs = int(n_envs)
        self.rng_mode = _lib.RNG_PHILOX if rng == "philox" else _lib.RNG_COMPAT
        if rng not in ("philox", "compat"):
            raise ValueError("rng must be 'philox' or 'compat'")
        h = C.c_void_p()
        opt = ChubOptions()
        opt.slot_kernel = _lib.SLOT_KERNELS[slot_kernel]
        opt.no_arena = int(bool(no_arena))
        opt.fused_step = _lib.FUSED_STEP[fused_step]  # "auto": one launch per step for small batches; "off" / "on" (parity cross-check)
        opt.tile = _lib.TILES[tile]  # packed slot kernel: "auto" (by working-set size), "small" (256 x 2) or "large" (512 x 4)
        opt.walk_ahead = _lib.WALK_AHEAD[walk_ahead]  # COMPAT batches: "auto" (step i + 1's stream walks beside step i's tails) or "off"
        opt.span_steps = 0 if span_steps == "auto" else (1 if span_steps == "off" else int(span_steps))  # chub_run_steps: steps per launch (one-launch handles)
        opt.span_tails = {"auto": 0, "same_wave": 1, "own_wave": 2}[span_tails]  # ... and the wave a span's tails run on (own_wave: a step behind the slots)
        opt.work_order = _lib.WORK_ORDER[work_order]  # PHILOX packed kernels: "auto" (XCD-aware while the streams are cache-resident) or "dispatch"
        self._copy_outputs = bool(copy_outputs)
        check(self._lib.chub_create_ex(C.byref(self.cfg), (data_dir or _lib.DATA_DIR).encode(), self.n_envs, int(env_id0),
                                       int(device), int(seed) & 0xFFFFFFFFFFFFFFFF, self.rng_mode, C.byref(opt), C.byref(h)))
        self._h = h
        self.obs_dim = self._lib.chub_obs_dim(h)
        self.act_dim = self._lib.chub_act_dim(h)
        self.n_slots = self.act_dim - 2
        self.piles = (int(station_list[0]), int(station_list[1]))
        self._device = int(device)
        self._host_allocs = []
        # the arrays the host-pointer entry points write into: pinned, so that the copies back are plain DMA
        self._obs = self._pinned_array((self.n_envs, self.obs_dim), np.float32)
        self._reward = self._pinned_array((self.n_envs,), np.float32)
        self._done = self._pinned_array((self.n_envs,), np.uint8)

    def _pinned_array(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(self._lib.chub_alloc_host(self._device, max(n, 1), C.byref(p)))
        self._host_allocs.append(p.value)
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        a[...] = 0
        return a

    def _out(self):
        if self._copy_outputs:
            return self._obs.copy(), self._reward.copy(), self._done.astype(bool), {}
        return self._obs, self._reward, self._done.view(np.bool_), {}

    # ---- hot path
    def reset(self, exo_days=None, exo_z=None):
        d = None if exo_days is None else np.ascontiguousarray(exo_days, dtype=np.int32).reshape(self.n_envs, 2)
        z = None if exo_z is None else np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)).reshape(self.n_envs, 3)
        check(self._lib.chub_reset(self._h, _ptr(d), _ptr(z), _ptr(self._obs)))
        return self._obs.copy() if self._copy_outputs else self._obs

    def step(self, actions, exo_z=None):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.n_envs, self.act_dim):  # MGR:148
            raise AssertionError("actions must have shape (%d, %d)" % (self.n_envs, self.act_dim))
        z = None if exo_z is None else np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)).reshape(self.n_envs, 3)
        check(self._lib.chub_step(self._h, _ptr(a), _ptr(z), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        return self._out()

    def pinned_actions(self):
        """the handle's pinned [N, A] f32 action buffer as a numpy array: fill it in place and pass it to step() to save the
        CPU copy into pinned memory (chub_host_actions)"""
        if getattr(self, "_pinned", None) is None:
            p = C.c_void_p()
            check(self._lib.chub_host_actions(self._h, C.byref(p)))
            buf = (C.c_float * (self.n_envs * self.act_dim)).from_address(p.value)
            self._pinned = np.frombuffer(buf, dtype=np.float32).reshape(self.n_envs, self.act_dim)
        return self._pinned

    # ---- packed actions (chub_step_bits): one bit per pile + the two tail floats, 16 bytes per env up to 64 piles
    @property
    def bit_words(self):
        return (self.n_slots + 63) // 64

    def pack_actions(self, actions, out=None):
        """[N, A] f32 action rows -> (pile_bits [N, ceil(S / 64)] u64, tail [N, 2] f32): exactly what action_to_real
        (evcssp_manager.py:384-393) keeps of them -- pile on iff f32((a + 1) / 2) >= 0.5, i.e. a >= -2^-25.  out: the arrays to
        fill (e.g. pinned_bits())"""
        a = np.asarray(actions, dtype=np.float32)
        if a.shape != (self.n_envs, self.act_dim):
            raise AssertionError("actions must have shape (%d, %d)" % (self.n_envs, self.act_dim))
        bits, tail = out if out is not None else (np.zeros((self.n_envs, self.bit_words), dtype=np.uint64),
                                                  np.zeros((self.n_envs, 2), dtype=np.float32))
        on = a[:, :self.n_slots] >= np.float32(-2.0 ** -25)
        packed = np.packbits(on, axis=1, bitorder="little")
        pad = self.bit_words * 8 - packed.shape[1]
        if pad:
            packed = np.concatenate([packed, np.zeros((self.n_envs, pad), dtype=np.uint8)], axis=1)
        bits[...] = np.ascontiguousarray(packed).view("<u8")
        tail[...] = a[:, self.n_slots:]
        return bits, tail

    def pinned_bits(self):
        """the handle's pinned (pile_bits, tail) staging arrays (chub_host_bits): fill them in place and pass them to step_bits()"""
        if getattr(self, "_pinned_bits", None) is None:
            pb, pt = C.c_void_p(), C.c_void_p()
            check(self._lib.chub_host_bits(self._h, C.byref(pb), C.byref(pt)))
            b = np.frombuffer((C.c_uint64 * (self.n_envs * self.bit_words)).from_address(pb.value), dtype=np.uint64)
            t = np.frombuffer((C.c_float * (self.n_envs * 2)).from_address(pt.value), dtype=np.float32)
            self._pinned_bits = (b.reshape(self.n_envs, self.bit_words), t.reshape(self.n_envs, 2))
        return self._pinned_bits

    def step_bits(self, pile_bits, tail, exo_z=None):
        """step() with the actions already reduced to what the env uses of them: 16 bytes per env over PCIe instead of 4 (S + 2)"""
        b = np.ascontiguousarray(pile_bits, dtype=np.uint64)
        t = np.ascontiguousarray(tail, dtype=np.float32)
        if b.shape != (self.n_envs, self.bit_words) or t.shape != (self.n_envs, 2):
            raise AssertionError("pile_bits must have shape (%d, %d) and tail (%d, 2)" % (self.n_envs, self.bit_words, self.n_envs))
        z = None if exo_z is None else np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)).reshape(self.n_envs, 3)
        check(self._lib.chub_step_bits(self._h, _ptr(b), _ptr(t), _ptr(z), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        return self._out()

    def load_actions(self, loads, tail):
        """[N, A] action array of the scalar-load mode: loads [N, 2] in kW (one per station), tail [N, 2] as in step()."""
        a = np.zeros((self.n_envs, self.act_dim), dtype=np.float32)
        loads = np.asarray(loads, dtype=np.float32).reshape(self.n_envs, 2)
        if self.piles[0] > 0:
            a[:, 0] = loads[:, 0]
        if self.piles[1] > 0:
            a[:, self.piles[0]] = loads[:, 1]
        a[:, self.n_slots:] = np.asarray(tail, dtype=np.float32).reshape(self.n_envs, 2)
        return a

    def step_load(self, loads, tail, exo_z=None):
        """Scalar-load control (the reference's evs_step(float), CHS.hpp:1169-1186 / 1480-1497): one kW target per station."""
        a = self.load_actions(loads, tail)
        z = None if exo_z is None else np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)).reshape(self.n_envs, 3)
        check(self._lib.chub_step_load(self._h, _ptr(a), _ptr(z), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        return self._out()

    # ---- tape mode (parity instrument, see include/chub.h): recorded decisions through the production kernels
    def tape_register_soc(self, soc):
        s = np.ascontiguousarray(soc, dtype=np.float32).ravel()
        ids = np.zeros(s.size, dtype=np.uint32)
        check(self._lib.chub_tape_register_soc(self._h, _ptr(s), int(s.size), _ptr(ids)))
        return ids

    def tape_clear_soc(self):
        check(self._lib.chub_tape_clear_soc(self._h))

    def set_slots(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.int32).reshape(self.n_envs, self.n_slots, 6)
        check(self._lib.chub_set_slots(self._h, _ptr(r)))

    def set_station_queue(self, line):
        q = np.ascontiguousarray(line, dtype=np.int32).reshape(self.n_envs, 2)
        check(self._lib.chub_set_station_queue(self._h, _ptr(q)))

    def reset_tape(self, occ_tape, car_tape, exo_days=None, exo_z=None):
        """evs_reset fed the reference's draws (chub_reset_tape): occ_tape [2, N] u32, car_tape [N, S, 2] u32; with exo_days [N, 2] and
        exo_z [N, 3] the tail takes renew_reset's days and make_state's normals from the caller too (chub_reset_tape_env)"""
        oc = np.ascontiguousarray(occ_tape, dtype=np.uint32).reshape(2, self.n_envs)
        ct = np.ascontiguousarray(car_tape, dtype=np.uint32).reshape(self.n_envs, self.n_slots, 2)
        if exo_days is None and exo_z is None:
            check(self._lib.chub_reset_tape(self._h, _ptr(oc), _ptr(ct), _ptr(self._obs)))
        else:
            if exo_days is None or exo_z is None:  # (the C API's own refusal, before numpy sees a None)
                raise ChubError("chub_reset_tape_env needs exo_days AND exo_z (the tail's side of the tape), or neither")
            d = np.ascontiguousarray(exo_days, dtype=np.int32).reshape(self.n_envs, 2)
            z = self._tape_normals(exo_z)
            check(self._lib.chub_reset_tape_env(self._h, _ptr(oc), _ptr(ct), _ptr(d), _ptr(z), _ptr(self._obs)))
        return self._obs.copy() if self._copy_outputs else self._obs

    def _tape_normals(self, exo_z):
        """[N, 3] f64 normals of a tape.  The wind process samples on every call (REN:66-69): its column must be finite.  The PV process samples only
        while the panel's table value is non-zero (REN:56-63: nothing is drawn at night) and the price noise on every fourth step (MGR:354):
        where the reference drew nothing its fixtures hold NaN, which becomes 0 here -- the tail does not look at a normal its process does not
        draw"""
        z = np.ascontiguousarray(exo_z, dtype=np.float64).reshape(self.n_envs, 3)
        if not np.isfinite(z[:, 1]).all():
            raise ChubError("tape normals: the wind column is drawn on every call and must be finite")
        return np.nan_to_num(z)

    def step_tape(self, actions, pk_tape, car_tape, exo_z=None, hv_tape=None):
        """one step from the tape (chub_step_tape); with exo_z [N, 3] f64 and hv_tape [N, W] u32 (word 0 = FCEV arrivals, word 1 + j =
        arrival j's SoC as f32 bits) the per-env tail replays the reference's draws as well (chub_step_tape_env)"""
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, self.act_dim)
        pk = np.ascontiguousarray(pk_tape, dtype=np.uint64).reshape(2, self.n_envs)
        ct = np.ascontiguousarray(car_tape, dtype=np.uint32).reshape(self.n_envs, self.n_slots, 2)
        if exo_z is None and hv_tape is None:
            check(self._lib.chub_step_tape(self._h, _ptr(a), _ptr(pk), _ptr(ct), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        else:
            if exo_z is None or hv_tape is None:
                raise ChubError("chub_step_tape_env needs exo_z AND hv_tape (the tail's side of the tape), or neither")
            z = self._tape_normals(exo_z)
            hv = np.ascontiguousarray(hv_tape, dtype=np.uint32)
            hv = hv.reshape(self.n_envs, hv.size // self.n_envs)
            check(self._lib.chub_step_tape_env(self._h, _ptr(a), _ptr(pk), _ptr(ct), _ptr(z), _ptr(hv), int(hv.shape[1]), _ptr(self._obs),
                                               _ptr(self._reward), _ptr(self._done)))
        return self._out()

    # ---- per-env clocks (every reference env owns its clock, MGR:137-140, 304-316): reset / step a subset of the envs
    def _mask(self, mask):
        m = np.ascontiguousarray(np.asarray(mask).astype(bool), dtype=np.uint8)
        if m.shape != (self.n_envs,):
            raise AssertionError("mask must have shape (%d,)" % self.n_envs)
        return m

    def reset_envs(self, mask, exo_days=None, exo_z=None):
        """reset the envs of `mask`; returns the current observation of every env ([N, D]: fresh rows for the envs of the
        mask, the others as their last call left them).  exo_days / exo_z: as for reset() (COMPAT handles)"""
        m = self._mask(mask)
        d = np.ascontiguousarray(exo_days, dtype=np.int32) if exo_days is not None else None
        z = np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)) if exo_z is not None else None
        check(self._lib.chub_reset_envs(self._h, _ptr(m), _ptr(d), _ptr(z), _ptr(self._obs)))
        return self._obs.copy()

    def step_envs(self, mask, actions, exo_z=None):
        """step the envs of `mask` only ([N, A] actions, the other rows are ignored); returns full-size obs, reward, done with
        the rows of the other envs as their last call left them"""
        m = self._mask(mask)
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.n_envs, self.act_dim):  # MGR:148
            raise AssertionError("actions must have shape (%d, %d)" % (self.n_envs, self.act_dim))
        z = np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)) if exo_z is not None else None
        check(self._lib.chub_step_envs(self._h, _ptr(m), _ptr(a), _ptr(z), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        return self._out()

    def step_load_envs(self, mask, loads, tail, exo_z=None):
        """step_load() for the envs of `mask` only"""
        m = self._mask(mask)
        a = self.load_actions(loads, tail)
        z = np.nan_to_num(np.ascontiguousarray(exo_z, dtype=np.float64)) if exo_z is not None else None
        check(self._lib.chub_step_load_envs(self._h, _ptr(m), _ptr(a), _ptr(z), _ptr(self._obs), _ptr(self._reward), _ptr(self._done)))
        return self._out()

    def env_clocks(self, ticks=False):
        """slot of day of every env (and, with ticks=True, the Philox tick of every env's last launch)"""
        t = np.zeros(self.n_envs, dtype=np.int32)
        tk = np.zeros(self.n_envs, dtype=np.uint32)
        check(self._lib.chub_env_clocks(self._h, _ptr(t), _ptr(tk)))
        return (t, tk) if ticks else t

    @property
    def clock_groups(self):
        """number of distinct env clocks right now (1 = lock-step)"""
        return self._lib.chub_clock_groups(self._h)

    # ---- device-pointer path (ints are raw device addresses, e.g. torch.Tensor.data_ptr())
    def reset_device(self, d_obs, d_exo_days=0, d_exo_z=0, stream=0):
        check(self._lib.chub_reset_device(self._h, d_exo_days or None, d_exo_z or None, d_obs, stream or None))

    def step_device(self, d_actions, d_obs, d_reward, d_done, d_exo_z=0, stream=0):
        check(self._lib.chub_step_device(self._h, d_actions, d_exo_z or None, d_obs, d_reward, d_done, stream or None))

    def step_device_packed(self, d_actions, d_packed, d_exo_z=0, stream=0):
        """One [N, D+2] f32 output buffer: obs, reward, done -- what a shard sends in the per-step RCCL gather."""
        check(self._lib.chub_step_device_packed(self._h, d_actions, d_exo_z or None, d_packed, stream or None))

    def step_bits_device(self, d_pile_bits, d_tail, d_obs, d_reward, d_done, d_exo_z=0, stream=0):
        """the device-pointer step fed one bit per pile ([N, bit_words] u64) + the two tail floats ([N, 2] f32) instead of action rows:
        on the packed slot kernel the step reads the bits themselves (chub_step_bits_device)"""
        check(self._lib.chub_step_bits_device(self._h, d_pile_bits, d_tail, d_exo_z or None, d_obs, d_reward, d_done, stream or None))

    def step_bits_device_packed(self, d_pile_bits, d_tail, d_packed, d_exo_z=0, stream=0):
        check(self._lib.chub_step_bits_device_packed(self._h, d_pile_bits, d_tail, d_exo_z or None, d_packed, stream or None))

    def random_actions_device(self, d_actions, key, batch, stream=0):
        check(self._lib.chub_random_actions_device(self._h, int(key), int(batch), d_actions, stream or None))

    def graph_begin(self, stream):
        """record the device-pointer calls issued on `stream` from here on instead of running them (chub_graph_begin)"""
        check(self._lib.chub_graph_begin(self._h, stream))

    def graph_end(self, stream):
        g = C.c_void_p()
        check(self._lib.chub_graph_end(self._h, stream, C.byref(g)))
        return g

    def graph_launch(self, graph, stream):
        check(self._lib.chub_graph_launch(graph, stream))

    def graph_destroy(self, graph):
        self._lib.chub_graph_destroy(graph)

    def profile_begin(self, max_steps, every=1):
        check(self._lib.chub_profile_begin(self._h, int(max_steps), int(every)))

    def profile_end(self):
        """-> (slot kernel ms summed, env kernel ms summed, steps covered)"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        check(self._lib.chub_profile_end(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def sync(self):
        check(self._lib.chub_sync(self._h))

    @property
    def uses_packed_kernel(self):
        return bool(self._lib.chub_uses_packed_kernel(self._h))

    @property
    def uses_fused_step(self):
        return bool(self._lib.chub_uses_fused_step(self._h))

    @property
    def uses_xcd_order(self):
        return bool(self._lib.chub_uses_xcd_order(self._h))

    @property
    def clock(self):
        return self._lib.chub_clock(self._h)

    # ---- introspection
    def set_telemetry(self, on=True):
        check(self._lib.chub_set_telemetry(self._h, int(bool(on))))
        self._tel_views = None

    def telemetry_views(self):
        """the handle's telemetry block as live numpy views, no copy and no device read (chub_telemetry_host): telem [T_COUNT, N]
        (one row per _lib.TELEMETRY_NAMES entry), obs64 [N, D], reward64 [N].  Current once the call that produced them has
        completed: reset() / step() return completed; after a device-pointer call, sync() first.  Handles of a few envs only (action rows
        of at most 16 KB in all: the drop-in class's one env): a batch keeps its block in device memory -- libchub returns
        CHUB_ERR_UNSUPPORTED here (ChubError) and telemetry() / obs_f64() / reward_f64() copy."""
        if getattr(self, "_tel_views", None) is None:
            pt, po, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(self._lib.chub_telemetry_host(self._h, C.byref(pt), C.byref(po), C.byref(pr)))
            N, D, T = self.n_envs, self.obs_dim, _lib.T_COUNT
            view = lambda p, n, shape: np.frombuffer((C.c_double * n).from_address(p.value), dtype=np.float64).reshape(shape)
            self._tel_views = (view(pt, T * N, (T, N)), view(po, N * D, (N, D)), view(pr, N, (N,)))
        return self._tel_views

    def slots(self):
        """list over stations of arrays [N, 9, piles_k]: car, charge, emergency, power, soc, init_soc, target_soc,
        stay_time, already_stay_time."""
        out = np.zeros((self.n_envs, 9 * self.n_slots), dtype=np.float32)
        check(self._lib.chub_get_slots(self._h, _ptr(out)))
        s0 = self.piles[0]
        a = out[:, :9 * s0].reshape(self.n_envs, 9, s0)
        b = out[:, 9 * s0:].reshape(self.n_envs, 9, self.piles[1])
        return [a, b]

    def station_scalars(self):
        out = np.zeros((self.n_envs, 2, 8), dtype=np.float64)
        check(self._lib.chub_get_station_scalars(self._h, _ptr(out)))
        return out

    def telemetry(self):
        out = np.zeros((self.n_envs, _lib.T_COUNT), dtype=np.float64)
        check(self._lib.chub_get_telemetry(self._h, _ptr(out)))
        return out

    def obs_f64(self):
        out = np.zeros((self.n_envs, self.obs_dim), dtype=np.float64)
        check(self._lib.chub_get_obs_f64(self._h, _ptr(out)))
        return out

    def reward_f64(self):
        out = np.zeros(self.n_envs, dtype=np.float64)
        check(self._lib.chub_get_reward_f64(self._h, _ptr(out)))
        return out

    def fcev_stuck_count(self):
        """envs whose FCEV forecourt is stuck: no prefix of its waiting list fits into 15 minutes any more, so -- as in the
        reference (HYD:270-276) -- nobody is served again until reset and the list only grows"""
        n = C.c_int64()
        check(self._lib.chub_fcev_stuck_count(self._h, C.byref(n)))
        return n.value

    def set_compat_seeds(self, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.uint32).reshape(self.n_envs, 2)
        check(self._lib.chub_set_rng_compat_seeds(self._h, _ptr(s)))

    def set_compat_state(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint32).reshape(self.n_envs, 33)
        check(self._lib.chub_set_rng_compat_state(self._h, _ptr(s)))

    def compat_state(self):
        out = np.zeros((self.n_envs, 33), dtype=np.uint32)
        check(self._lib.chub_get_rng_compat_state(self._h, _ptr(out)))
        return out

    def compat_replay_constructor(self):
        check(self._lib.chub_compat_replay_constructor(self._h))

    def set_ou_state(self, ou):
        o = np.ascontiguousarray(ou, dtype=np.float64).reshape(self.n_envs, 3)
        check(self._lib.chub_set_ou_state(self._h, _ptr(o)))

    def get_state(self):
        """Snapshot of the whole simulation state as bytes (see chub_get_state)."""
        n = self._lib.chub_state_size(self._h)
        if n < 0:
            check(int(n))
        buf = np.empty(n, dtype=np.uint8)
        check(self._lib.chub_get_state(self._h, _ptr(buf), n))
        return buf

    def set_state(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.chub_set_state(self._h, _ptr(b), b.size))

    def hy_table(self, env=None):
        """hy_power_speed_list (HYD:154-157): the handle's table, or env i's own (COMPAT after compat_replay_constructor)"""
        out = np.zeros(102, dtype=np.float64)
        if env is None:
            check(self._lib.chub_get_hy_table(self._h, _ptr(out)))
        else:
            check(self._lib.chub_get_hy_table_env(self._h, int(env), _ptr(out)))
        return out

    def set_hy_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        if t.shape != (102,):
            raise ValueError("table must have 102 entries")
        check(self._lib.chub_set_hy_table(self._h, _ptr(t)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.chub_destroy(self._h)
            self._h = None
            self._obs = self._reward = self._done = self._pinned = self._pinned_bits = self._tel_views = None
            for p in self._host_allocs:
                self._lib.chub_free_host(self._device, p)
            self._host_allocs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["VecChargingHub", "make_config", "ChubError"]
