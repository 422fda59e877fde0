"""ctypes bindings for the test-side libraries (TEST INFRASTRUCTURE).

* ``orc``  -> oracle/liboracle.so   : the CPU restatement (plain C); ``big_oracle()`` -> liboracle_big.so
* ``ref()`` -> oracle/_ref/libchs_ref.so : the real reference C++ core compiled from
  /root/reference (present only where it was built; its static initialiser reads
  ``car_flow_possibility_list_save.csv`` from the CWD, CHS.hpp:175, so we chdir for the load).
"""
import contextlib
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DATA_DIR = os.path.join(ROOT, "charginghub-env_amd", "data")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

FAST, SLOW = 0, 1
COMPAT, PHILOX = 0, 1
PU = dict(ARRIVE=1, INIT=2, RENEGE=3, BALK=4, SOC=5, TGT=6, LATE=7, HV=8, HVSOC=9, OU=10, DAY=11)


class OrcConfig(C.Structure):
    _fields_ = [("piles", C.c_int * 2), ("type", C.c_int * 2), ("constant_charging", C.c_int),
                ("hydro_prod_rate", C.c_double), ("hydro_store_vlt", C.c_double), ("init_soc", C.c_double),
                ("fc_max_power", C.c_double), ("fcev_permeate", C.c_double), ("renew_fluctuate", C.c_double),
                ("price_fluctuate", C.c_double), ("hydro_loss", C.c_double)]


def make_config(piles=(20, 25), types=("fast", "slow"), constant_charging=False, hydro_prod_rate=100.0,
                hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01,
                renew_fluctuate=0.0, price_fluctuate=0.0, hydro_loss=0.0):
    c = OrcConfig()
    c.piles[0], c.piles[1] = piles
    c.type[0] = FAST if types[0] == "fast" else SLOW
    c.type[1] = FAST if types[1] == "fast" else SLOW
    c.constant_charging = int(constant_charging)
    c.hydro_prod_rate = hydro_prod_rate
    c.hydro_store_vlt = hydro_store_vlt
    c.init_soc = init_soc
    c.fc_max_power = fc_max_power
    c.fcev_permeate = fcev_permeate
    c.renew_fluctuate = renew_fluctuate
    c.price_fluctuate = price_fluctuate
    c.hydro_loss = hydro_loss
    return c


def _build_oracle(name="liboracle.so"):
    so = os.path.join(ORACLE_DIR, name)
    srcs = [os.path.join(ORACLE_DIR, "chub_oracle.c"), os.path.join(ORACLE_DIR, "chub_oracle.h")]
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return so


def _load_oracle(name="liboracle.so"):
    lib = C.CDLL(_build_oracle(name))
    P, I, D, F = C.c_void_p, C.c_int, C.c_double, C.c_float
    sig = {
        "orc_parse_float": (F, [C.c_char_p, I]),
        "orc_tables_load": (P, [C.c_char_p]),
        "orc_tables_free": (None, [P]),
        "orc_tables_cdf": (F, [P, I, I]),
        "orc_arrival_index": (I, [P, I, I]),
        "orc_count_fast": (I, [I]),
        "orc_count_slow": (I, [I]),
        "orc_count_hv": (I, [I, F, F]),
        "orc_uniform_level": (F, [I, F, F]),
        "orc_rng_alloc": (P, []),
        "orc_rng_free": (None, [P]),
        "orc_rng_set_tick": (None, [P, C.c_uint32]),
        "orc_rng_seed_compat": (None, [P, C.c_uint32, C.c_uint32]),
        "orc_rng_seed_philox": (None, [P, C.c_uint64, C.c_uint32]),
        "orc_glibc_rand": (C.c_uint32, [P]),
        "orc_minstd_next": (C.c_uint32, [P]),
        "orc_philox4x32_10": (None, [P, P, P]),
        "orc_draw_k": (I, [P, I, I, I]),
        "orc_mk_soc": (F, [P]),
        "orc_mk_late_time": (I, [P]),
        "orc_soc_from_word": (F, [P, C.c_uint32]),
        "orc_soc_level_from_word": (F, [P, C.c_uint32]),
        "orc_soc_level_value": (F, [P, C.c_uint32]),
        "orc_late_from_word": (I, [P, C.c_uint32]),
        "orc_init_station_car_number": (I, [P, P, I, I]),
        "orc_normal_from_word": (F, [P, C.c_uint32]),
        "orc_rng_export_glibc128": (None, [P, P]),
        "orc_rng_import_glibc128": (None, [P, P]),
        "orc_curve_slow": (F, [I, F, I]),
        "orc_curve_fast": (F, [I, F, I]),
        "orc_station_alloc": (P, []),
        "orc_station_free": (None, [P]),
        "orc_station_init": (None, [P, I, I, I, I, I, I]),
        "orc_station_reset": (None, [P, P, P]),
        "orc_station_step": (None, [P, P, P, P]),
        "orc_station_step_load": (None, [P, P, P, F]),
        "orc_station_scalars": (None, [P, P]),
        "orc_station_slots": (None, [P, P]),
        "orc_j2601_target_pressure": (D, [D]),
        "orc_j2601_time_mass": (None, [D, P, P]),
        "orc_electrolyser_power": (D, [D, C.c_long]),
        "orc_electrolyser_cells": (C.c_long, [D]),
        "orc_compressor_kw": (D, [D]),
        "orc_env_alloc": (P, []),
        "orc_env_free": (None, [P]),
        "orc_env_init": (None, [P, P, P]),
        "orc_env_init_compat_ctor": (None, [P, P, P, C.c_uint32, C.c_uint32]),
        "orc_env_reset": (None, [P, P, P, P]),
        "orc_env_step": (None, [P, P, P, P, P, P]),
        "orc_env_obs_dim": (I, [P]),
        "orc_env_station": (P, [P, I]),
        "orc_env_rng": (P, [P]),
        "orc_env_hy_table": (None, [P, P]),
        "orc_env_set_hy_table": (None, [P, P]),
        "orc_env_telemetry": (I, [P, P]),
        "orc_env_q_overflow": (I, [P]),
        "orc_vec_create": (P, [P, P, C.c_long, C.c_long, I, C.c_uint64]),
        "orc_vec_destroy": (None, [P]),
        "orc_vec_env": (P, [P, C.c_long]),
        "orc_vec_reset": (None, [P, P, P, P]),
        "orc_vec_step": (None, [P, P, P, P, P, P, I]),
        "orc_vec_step_load": (None, [P, P, P, P, P, P, I]),
        "orc_env_step_load": (None, [P, P, P, P, P, P]),
        "orc_check_canon_division": (C.c_long, [C.c_long, C.c_uint64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


orc = _load_oracle()
_tables = None
_big = None


@contextlib.contextmanager
def big_oracle(*modules):
    """Stations of more than 256 piles: inside the block `orclib.orc` -- and the name `orc` of every module given -- is
    liboracle_big.so, the same source compiled with room for 4096 piles per station (ORC_MAX_PILES).  The tables are plain data
    and shared."""
    global orc, _big
    if _big is None:
        _big = _load_oracle("liboracle_big.so")
    me = sys.modules[__name__]
    saved = [(m, m.orc) for m in (me,) + tuple(modules)]
    try:
        for m, _ in saved:
            m.orc = _big
        yield _big
    finally:
        for m, o in saved:
            m.orc = o


def tables():
    global _tables
    if _tables is None:
        _tables = orc.orc_tables_load(DATA_DIR.encode())
        assert _tables, "oracle could not load data tables from " + DATA_DIR
    return _tables


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_ref = None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libchs_ref.so"))


def ref():
    """The real reference core (CHS.hpp) behind oracle/ref_driver.cpp."""
    global _ref
    if _ref is None:
        cwd = os.getcwd()
        os.chdir(DATA_DIR)  # CHS.hpp:175 reads the CSV from the CWD at load time
        try:
            lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libchs_ref.so"))
        finally:
            os.chdir(cwd)
        P, I, D, F, U = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_uint
        sig = {
            "ref_seed": (None, [U, U]), "ref_rng_state_size": (I, []), "ref_rng_save": (None, [P]),
            "ref_rng_load": (None, [P]), "ref_c_rand": (I, []), "ref_minstd_next": (U, []),
            "ref_uniform_rand": (F, [F, F]), "ref_mk_soc": (F, []), "ref_mk_late_time": (I, [I]),
            "ref_init_station_car_number": (I, [I]), "ref_cdf_rows": (I, []), "ref_cdf_cols": (I, [I]),
            "ref_cdf": (D, [I, I]), "ref_give_car_number": (I, [I]), "ref_ev_fast": (I, [I]),
            "ref_ev_slow": (I, [I]), "ref_hv": (I, [I, F, F]), "ref_parse_float": (F, [C.c_char_p]),
            "ref_curve_slow": (F, [I, F, I]), "ref_curve_fast": (F, [I, F, I]),
            "ref_station_new": (P, [I, I, I, I]), "ref_station_free": (None, [P]),
            "ref_station_reset": (None, [P]), "ref_station_step": (None, [P, P, I]),
            "ref_station_step_load": (None, [P, F]), "ref_station_scalars": (None, [P, P]),
            "ref_station_slots": (None, [P, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        assert lib.ref_cdf_rows() == 96, "reference CSV did not load (CWD)"
        _ref = lib
    return _ref


class OrcStation:
    """One oracle station + its own RNG (mirrors ref_driver's handle API)."""

    def __init__(self, typ, piles, wait=True, constant_charging=False, index=0, slot_base=0):
        self.n = piles
        self.s = orc.orc_station_alloc()
        self.r = orc.orc_rng_alloc()
        orc.orc_station_init(self.s, typ, piles, int(wait), int(constant_charging), index, slot_base)

    def seed_compat(self, g, m):
        orc.orc_rng_seed_compat(self.r, g, m)

    def reset(self):
        orc.orc_station_reset(self.s, self.r, tables())

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        orc.orc_station_step(self.s, self.r, tables(), ptr(a))

    def step_load(self, load):
        orc.orc_station_step_load(self.s, self.r, tables(), float(load))

    def scalars(self):
        out = np.zeros(8)
        orc.orc_station_scalars(self.s, ptr(out))
        return out

    def slots(self):
        out = np.zeros((9, self.n), dtype=np.float32)
        orc.orc_station_slots(self.s, ptr(out))
        return out

    def __del__(self):
        try:
            orc.orc_station_free(self.s)
            orc.orc_rng_free(self.r)
        except Exception:
            pass


class RefStation:
    def __init__(self, typ, piles, wait=True, constant_charging=False):
        self.n = piles
        self.h = ref().ref_station_new(typ, piles, int(wait), int(constant_charging))

    def reset(self):
        ref().ref_station_reset(self.h)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        ref().ref_station_step(self.h, ptr(a), len(a))

    def step_load(self, load):
        ref().ref_station_step_load(self.h, float(load))

    def scalars(self):
        out = np.zeros(8)
        ref().ref_station_scalars(self.h, ptr(out))
        return out

    def slots(self):
        out = np.zeros((9, self.n), dtype=np.float32)
        ref().ref_station_slots(self.h, ptr(out))
        return out

    def __del__(self):
        try:
            ref().ref_station_free(self.h)
        except Exception:
            pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def golden_config(g):
    """orc_config from a fixture's recorded constructor kwargs."""
    sl = [int(x) for x in g["kw_station_list"]]
    st = ["fast" if int(x) == 0 else "slow" for x in g["kw_station_type"]]
    return make_config(piles=sl, types=st, constant_charging=bool(g["kw_constant_charging"]),
                       hydro_prod_rate=float(g["kw_hydro_prod_rate"]), hydro_store_vlt=float(g["kw_hydro_store_vlt"]),
                       init_soc=float(g["kw_init_soc"]), fc_max_power=float(g["kw_fc_max_power"]),
                       fcev_permeate=float(g["kw_fcev_permeate"]), renew_fluctuate=float(g["kw_renew_fluctuate"]),
                       price_fluctuate=float(g["kw_price_fluctuate"]), hydro_loss=float(g["kw_hydro_loss"]))


GOLDEN_ENV = ["env_c1_envtest", "env_c3_random", "env_c2_random", "env_c5_random", "env_slow_only_fcev",
              "env_clamp", "env_full_tank", "env_constant", "env_fcev_queue", "env_small_fast_neg", "env_fcev_queue_deep",
              "env_big_100_70",
              # round 4, second batch: two stations of the same kind, no electrolyser, a permeability above 1, one pile per station,
              # constant-power fleet on swapped station kinds
              "env_slow_slow", "env_fast_fast", "env_no_electrolyser", "env_permeate_cap", "env_one_pile", "env_constant_swapped",
              # round 5: stepping past `done` without a reset (MGR:271-299; the registered horizon is 999 steps, evcssp_env_cpp/__init__.py:6):
              # one episode of 250 steps (C3 hub, fluctuating series, tank loss) and one of 200 (C2 hub)
              "env_past_done", "env_past_done_c2",
              # ... and every kwarg the reference gives a default left to it (a 430 m^3/h electrolyser, a 5000 m^3 tank at SOC 0.5, 100 fuel cells)
              "env_defaults",
              # the tank at its 10 % floor from the first step on (unmet forecourt demand) and full to the brim (the electrolyser idles)
              "env_tank_floor", "env_tank_brim"]


# round 6: stations of more than 256 piles (300 fast + 270 slow) -- the oracle side of these runs on liboracle_big.so (big_oracle())
GOLDEN_ENV_BIG = ["env_big_300_270"]


class OrcEnv:
    def __init__(self, cfg, ctor_seeds=None):
        self.cfg = cfg
        self.e = orc.orc_env_alloc()
        if ctor_seeds is None:
            orc.orc_env_init(self.e, C.byref(cfg), tables())
        else:
            orc.orc_env_init_compat_ctor(self.e, C.byref(cfg), tables(), ctor_seeds[0], ctor_seeds[1])
        self.D = orc.orc_env_obs_dim(C.byref(cfg))
        self.S = cfg.piles[0] + cfg.piles[1]

    def seed_compat(self, g, m):
        orc.orc_rng_seed_compat(orc.orc_env_rng(self.e), g, m)

    def seed_philox(self, seed, env_id):
        orc.orc_rng_seed_philox(orc.orc_env_rng(self.e), seed, env_id)

    def reset(self, days=None, z=None):
        obs = np.zeros(self.D)
        d = np.asarray(days, dtype=np.int32) if days is not None else None
        zz = np.nan_to_num(np.asarray(z, dtype=np.float64)) if z is not None else None
        orc.orc_env_reset(self.e, ptr(d), ptr(zz), ptr(obs))
        return obs

    def step(self, action, z=None):
        obs = np.zeros(self.D)
        r = C.c_double()
        d = C.c_int()
        a = np.ascontiguousarray(action, dtype=np.float32)
        zz = np.nan_to_num(np.asarray(z, dtype=np.float64)) if z is not None else None
        orc.orc_env_step(self.e, ptr(a), ptr(zz), ptr(obs), C.byref(r), C.byref(d))
        return obs, r.value, bool(d.value)

    def telemetry(self):
        out = np.zeros(38)
        orc.orc_env_telemetry(self.e, ptr(out))
        return out

    def q_overflow(self):
        return orc.orc_env_q_overflow(self.e)

    def hy_table(self):
        out = np.zeros(102)
        orc.orc_env_hy_table(self.e, ptr(out))
        return out

    def set_hy_table(self, tab):
        t = np.ascontiguousarray(tab, dtype=np.float64)
        assert t.shape == (102,)
        orc.orc_env_set_hy_table(self.e, ptr(t))

    def station_scalars(self, k):
        out = np.zeros(8)
        orc.orc_station_scalars(orc.orc_env_station(self.e, k), ptr(out))
        return out

    def station_slots(self, k):
        n = self.cfg.piles[k]
        out = np.zeros((9, n), dtype=np.float32)
        orc.orc_station_slots(orc.orc_env_station(self.e, k), ptr(out))
        return out

    def __del__(self):
        try:
            orc.orc_env_free(self.e)
        except Exception:
            pass
