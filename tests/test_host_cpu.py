"""CPU-side tests (no GPU): the C-ABI library loads and exports what include/chub.h declares, argument / data
errors surface as error codes, the host mirror of the reference interface behaves, the PHILOX-mode sampling tables
have the reference's distributions, and the oracle's Philox streams do not depend on the sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orclib
from orclib import orc, ptr

ROOT = orclib.ROOT


def chub():
    import charginghub_env_amd as m
    return m


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "chub.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(chub_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    m = chub()
    lib = C.CDLL(m.lib_path())
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libchub.so does not export %s declared in include/chub.h" % n
    from charginghub_env_amd import _lib
    assert sorted(_lib.EXPORTED) == names, "ctypes binding out of sync with include/chub.h"


def test_telemetry_columns_and_option_struct_match_the_header():
    """the CHUB_T_* enum of include/chub.h, the device-side column count and the Python names are one list; chub_options is the
    same 8 ints on both sides"""
    from charginghub_env_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "chub.h")).read()
    body = hdr[hdr.index("CHUB_T_HY_ACT = 0"):hdr.index("CHUB_T_COUNT")]
    cols = re.findall(r"\bCHUB_T_[A-Z0-9_]+", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert len(cols) == _lib.T_COUNT == len(_lib.TELEMETRY_NAMES) == len(set(_lib.TELEMETRY_NAMES))
    dev = open(os.path.join(ROOT, "charginghub-env_amd", "csrc", "chub_device.h")).read()
    assert int(re.search(r"kTelemCount = (\d+);", dev).group(1)) == _lib.T_COUNT
    assert cols[_lib.T["Store_SOC"]] == "CHUB_T_STORE_SOC" and cols[_lib.T["price_now"]] == "CHUB_T_PRICE_NOW"
    assert cols[_lib.T["flow_in_1"]] == "CHUB_T_FLOW1" and cols[-1] == "CHUB_T_FLOW1"
    opt = hdr[hdr.index("typedef struct chub_options"):hdr.index("} chub_options;")]
    fields = re.findall(r"int32_t (\w+)(?:\[(\d+)\])?;", opt)
    assert [(n, int(k or 1)) for n, k in fields] == [(n, getattr(t, "_length_", 1)) for n, t in _lib.ChubOptions._fields_]
    assert C.sizeof(_lib.ChubOptions) == 32


def test_option_values_are_validated_before_any_device_call():
    """chub_create_ex refuses out-of-range chub_options with CHUB_ERR_ARG and a message naming the field, GPU or not (the option checks come
    before the device is looked for)"""
    m = chub()
    lib = m.load_library()
    from charginghub_env_amd import _lib
    good = m.make_config([20, 25], ["fast", "slow"])
    data = _lib.DATA_DIR.encode()
    for field, bad in (("slot_kernel", 3), ("fused_step", -1), ("tile", 3), ("walk_ahead", 2), ("work_order", 2), ("span_steps", 97), ("span_steps", -1),
                       ("span_tails", 3), ("span_tails", -1)):
        opt = _lib.ChubOptions()
        setattr(opt, field, bad)
        h = C.c_void_p()
        assert lib.chub_create_ex(C.byref(good), data, 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(opt), C.byref(h)) == -1, field
        assert ("chub_options." + field).encode() in lib.chub_last_error(), (field, lib.chub_last_error())
        assert not h.value


def test_create_fails_loudly_without_gpu_and_validates_arguments():
    m = chub()
    lib = m.load_library()
    from charginghub_env_amd import _lib
    h = C.c_void_p()
    good = m.make_config([20, 25], ["fast", "slow"])
    data = _lib.DATA_DIR.encode()
    if lib.chub_device_count() == 0:
        rc = lib.chub_create(C.byref(good), data, 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h))
        assert rc == -3 and b"no HIP device" in lib.chub_last_error()  # CHUB_ERR_HIP: no CPU fallback
        with pytest.raises(m.ChubError):
            m.VecChargingHub(4, [20, 25], ["fast", "slow"])
    # argument errors come first, device or not
    assert lib.chub_create(C.byref(good), data, 0, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h)) == -1
    assert lib.chub_create(C.byref(good), data, 4, 0, 0, 1, 7, C.byref(h)) == -1
    bad = m.make_config([0, 0], ["fast", "slow"])
    assert lib.chub_create(C.byref(bad), data, 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h)) == -1
    assert b"must have fast pile or slow pile" in lib.chub_last_error()  # MGR:336
    big = m.make_config([4097, 1], ["fast", "slow"])  # stations of up to 4096 piles (the reference: any count, CHS.hpp:1148, 1458)
    assert lib.chub_create(C.byref(big), data, 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h)) == -4
    assert b"4096 piles" in lib.chub_last_error()
    big = m.make_config([65, 4097], ["fast", "slow"])  # ... in both RNG modes
    assert lib.chub_create(C.byref(big), data, 4, 0, 0, 1, _lib.RNG_COMPAT, C.byref(h)) == -4
    lowsoc = m.make_config([4, 4], ["fast", "slow"], init_soc=0.05)
    assert lib.chub_create(C.byref(lowsoc), data, 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h)) == -1  # HYD:137
    # missing data directory -> CHUB_ERR_DATA (the reference prints "File not found" and carries on, CHS.hpp:102-105)
    assert lib.chub_create(C.byref(good), b"/nonexistent", 4, 0, 0, 1, _lib.RNG_PHILOX, C.byref(h)) == -2
    assert lib.chub_destroy(None) == 0


def test_make_config_mirrors_reference_kwargs():
    m = chub()
    c = m.make_config([20, 25], ["fast", "slow"])
    assert (c.hydro_prod_rate, c.hydro_store_vlt, c.fc_max_power, c.init_soc, c.fcev_permeate) == (430.0, 5000.0, 100.0, 0.5, 0.01)
    with pytest.raises(ValueError):  # AGG:196
        m.make_config([1, 1], ["fast", "medium"])
    with pytest.raises(AssertionError):  # MGR:37
        m.make_config([1, 1, 1], ["fast", "slow", "slow"])
    b = m.Box(-1.0, 1.0, (47,))
    assert b.shape == (47,) and b.contains(b.sample())


def test_shard_range():
    from charginghub_env_amd.sharded import shard_range
    assert [shard_range(65536, 8, r) for r in (0, 1, 7)] == [(0, 8192), (8192, 8192), (57344, 8192)]
    with pytest.raises(ValueError):
        shard_range(10, 3, 0)


def test_philox_sampling_tables_match_reference_distributions():
    """PHILOX mode draws mk_soc / mk_late_time / N(0,1) from tabulated inverse CDFs; their distributions must be
    the ones the reference's streams produce (CHS.hpp:804-830)."""
    t = orclib.tables()
    rs = np.random.RandomState(0)
    w = rs.randint(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32)
    soc = np.array([orc.orc_soc_from_word(t, int(x)) for x in w[:100000]])
    late = np.array([orc.orc_late_from_word(t, int(x)) for x in w[:100000]])
    z = np.array([orc.orc_normal_from_word(t, int(x)) for x in w])
    g = orc.orc_rng_alloc()
    orc.orc_rng_seed_compat(g, 5, 6)
    rsoc = np.array([orc.orc_mk_soc(g) for _ in range(100000)])
    rlate = np.array([orc.orc_mk_late_time(g) for _ in range(100000)])
    orc.orc_rng_free(g)
    lsoc = np.array([orc.orc_soc_level_from_word(t, int(x)) for x in w[:100000]])   # EV arrivals: 2048 levels
    qs = [0.05, 0.25, 0.5, 0.75, 0.9]
    for got in (soc, lsoc):
        assert got.min() >= 25.0 and got.max() <= 70.0
        assert np.allclose(np.quantile(got, qs), np.quantile(rsoc, qs), atol=0.35)
        assert abs((got == 25.0).mean() - (rsoc == 25.0).mean()) < 0.005   # clip mass at driver_experience = 10
        assert abs((got == 70.0).mean() - (rsoc == 70.0).mean()) < 0.005   # ... and = 1
    lv = np.array([orc.orc_soc_level_value(t, l) for l in range(2048)])
    assert np.all(np.diff(lv) <= 0) and lv[0] == 70.0 and lv[-1] == 25.0   # monotone in the level, clipped ends
    assert np.abs(lsoc - soc).max() < 0.1                                  # level = the continuous variate, quantised
    for j in range(8):
        assert abs((late == j).mean() - (rlate == j).mean()) < 0.006, j
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((np.abs(z) > 1.96).mean() - 0.05) < 0.003 and abs((z > 3.0).mean() - 0.00135) < 0.0005
    # exact symmetry / monotonicity of the two-level table
    assert orc.orc_normal_from_word(t, 0) == -orc.orc_normal_from_word(t, 0xFFFFFFFF)
    ws = np.sort(w[:5000])
    zs = np.array([orc.orc_normal_from_word(t, int(x)) for x in ws])
    assert np.all(np.diff(zs) >= 0)


def test_oracle_philox_streams_are_shard_independent():
    cfg = orclib.make_config(piles=(20, 25), types=("fast", "slow"))
    n = 24
    whole = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 100, orclib.PHILOX, 77)
    a = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n // 2, 100, orclib.PHILOX, 77)
    b = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n // 2, 100 + n // 2, orclib.PHILOX, 77)
    D, A = 13, 47
    ow, oa, ob = np.zeros((n, D)), np.zeros((n // 2, D)), np.zeros((n // 2, D))
    rw, ra, rb = np.zeros(n), np.zeros(n // 2), np.zeros(n // 2)
    dw, da, db = (np.zeros(k, dtype=np.uint8) for k in (n, n // 2, n // 2))
    orc.orc_vec_reset(whole, None, None, ptr(ow))
    orc.orc_vec_reset(a, None, None, ptr(oa))
    orc.orc_vec_reset(b, None, None, ptr(ob))
    assert np.array_equal(ow, np.concatenate([oa, ob]))
    rs = np.random.RandomState(1)
    for t in range(100):
        act = rs.uniform(-1, 1, size=(n, A)).astype(np.float32)
        a0, a1 = np.ascontiguousarray(act[:n // 2]), np.ascontiguousarray(act[n // 2:])
        orc.orc_vec_step(whole, ptr(act), None, ptr(ow), ptr(rw), ptr(dw), 3)
        orc.orc_vec_step(a, ptr(a0), None, ptr(oa), ptr(ra), ptr(da), 1)
        orc.orc_vec_step(b, ptr(a1), None, ptr(ob), ptr(rb), ptr(db), 2)
        assert np.array_equal(ow, np.concatenate([oa, ob])) and np.array_equal(rw, np.concatenate([ra, rb]))
        assert dw.all() == (t == 95)
        if t == 95:
            for h, o in ((whole, ow), (a, oa), (b, ob)):
                orc.orc_vec_reset(h, None, None, ptr(o))
    for h in (whole, a, b):
        orc.orc_vec_destroy(h)


def test_philox_mode_is_statistically_the_reference_process():
    """The production RNG mode draws from different streams (Philox + tabulated inverse CDFs) than the reference
    (glibc rand + minstd + polar normals).  It must still simulate the same stochastic process: compare episode-level
    statistics of the two modes of the oracle over many envs under the same policy."""
    cfg = orclib.make_config(piles=(20, 25), types=("fast", "slow"), fcev_permeate=0.05)
    n, D, A = 384, 13, 47
    rs = np.random.RandomState(3)
    stats = {}
    for mode in (orclib.COMPAT, orclib.PHILOX):
        h = orc.orc_vec_create(C.byref(cfg), orclib.tables(), n, 0, mode, 2024)
        obs, rew, done = np.zeros((n, D)), np.zeros(n), np.zeros(n, dtype=np.uint8)
        rs_a = np.random.RandomState(11)
        days = np.stack([rs.randint(0, 100, n), rs.randint(0, 150, n)], axis=1).astype(np.int32)
        ret = np.zeros(n)
        cars, line, flow, soc_new = [], [], [], []
        if mode == orclib.COMPAT:
            orc.orc_vec_reset(h, ptr(days), ptr(rs.normal(size=(n, 3))), ptr(obs))
        else:
            orc.orc_vec_reset(h, None, None, ptr(obs))
        for t in range(96):
            act = rs_a.uniform(-1, 1, size=(n, A)).astype(np.float32)
            z = rs.normal(size=(n, 3)) if mode == orclib.COMPAT else None
            orc.orc_vec_step(h, ptr(act), ptr(z) if z is not None else None, ptr(obs), ptr(rew), ptr(done), 4)
            ret += rew
            for e in range(0, n, 8):
                for k in (0, 1):
                    sc = np.zeros(8)
                    orc.orc_station_scalars(orc.orc_env_station(orc.orc_vec_env(h, e), k), ptr(sc))
                    cars.append(sc[3]); line.append(sc[4]); flow.append(sc[5])
        stats[mode] = dict(ret=ret, cars=np.array(cars), line=np.array(line), flow=np.array(flow))
        orc.orc_vec_destroy(h)
    a, b = stats[orclib.COMPAT], stats[orclib.PHILOX]
    # PV / wind days and OU noise differ between the two runs as well, so compare with sampling error in mind
    se = np.sqrt(a["ret"].var() / n + b["ret"].var() / n)
    assert abs(a["ret"].mean() - b["ret"].mean()) < 5 * se + 0.5, (a["ret"].mean(), b["ret"].mean(), se)
    for key, tol in (("cars", 0.35), ("line", 0.25), ("flow", 0.15)):
        assert abs(a[key].mean() - b[key].mean()) < tol, (key, a[key].mean(), b[key].mean())
    assert abs(a["cars"].std() - b["cars"].std()) < 0.4


def test_canonical_division_by_the_constant_is_the_division_itself():
    """libstdc++'s generate_canonical<double> divides its two-draw sum by R * R (R = 2^31 - 2); the device's stream walk (CompatStreamT::canon_d,
    csrc/chub_kernels.hip) multiplies by the rounded reciprocal and corrects with two fmas.  The identity is a theorem (Markstein); here it is
    also exercised on 4e8 dividends of the walk's own shape, incl. the smallest and largest second draws."""
    import orclib
    assert orclib.orc.orc_check_canon_division(400_000_000, 20260105) == 0
