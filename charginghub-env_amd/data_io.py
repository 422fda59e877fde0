"""User-supplied input series for the hub: arrival CDFs, price, PV and wind profiles.

The reference reads four assets next to its sources: ``car_flow_possibility_list_save.csv`` (96 x 301 arrival CDFs,
CHS.hpp:93-175), ``price_after_MAD_96.pkl`` (96 prices, Aggregator_Simple.py:9-15), ``pv_power_100.pkl`` (100 days x 96,
renewable.py:10-13) and ``wd_power_150.pkl`` (150 days x 96, renewable.py:15-18).  libchub takes the same four tables from
a *data directory* (``chub_create(..., data_dir, ...)``): the CSV unchanged plus three flat little-endian float64 files.

``write_data_dir`` builds such a directory from arrays or files (``.csv``, ``.npy``, flat ``.f64``/``.f32`` binary, or the
reference's own ``.pkl`` lists); whatever is not given is taken from the packaged defaults.  The day counts 100 / 150 are
part of the reference's contract (``random.randint(0, 99)`` / ``(0, 149)``, renewable.py:51-53): shorter series are
repeated cyclically to that length, longer ones are rejected.
"""
import os
import shutil

import numpy as np

from . import _lib

PV_DAYS, WD_DAYS, SLOTS, CDF_COLS = 100, 150, 96, 301
_CDF = "car_flow_possibility_list_save.csv"
_STATIC = ["soc_d_icdf_4097.f32", "late_thr_16.u32", "normal_icdf_4097.f32", "normal_tail_4097.f32"]


def load_series(src, dtype=np.float64):
    """array-like, or a path ending in .npy / .csv / .txt / .f64 / .f32 / .pkl -> ndarray (pickles: trusted files only)"""
    if isinstance(src, (str, os.PathLike)):
        path = os.fspath(src)
        ext = os.path.splitext(path)[1].lower()
        if ext == ".npy":
            a = np.load(path, allow_pickle=False)
        elif ext in (".csv", ".txt"):
            a = np.loadtxt(path, delimiter=",", ndmin=1)
        elif ext == ".f64":
            a = np.fromfile(path, dtype="<f8")
        elif ext == ".f32":
            a = np.fromfile(path, dtype="<f4")
        elif ext == ".pkl":
            import pickle

            with open(path, "rb") as f:
                a = pickle.load(f)
        else:
            raise ValueError("unsupported series file type: %s" % path)
    else:
        a = src
    a = np.asarray(a, dtype=dtype)
    if not np.all(np.isfinite(a)):
        raise ValueError("series contains non-finite values")
    return a


def _days(a, days, name):
    a = np.atleast_2d(a)
    if a.ndim != 2 or a.shape[1] != SLOTS:
        if a.size % SLOTS == 0 and a.ndim <= 2:
            a = a.reshape(-1, SLOTS)
        else:
            raise ValueError("%s must have %d values per day, got shape %s" % (name, SLOTS, a.shape))
    if a.shape[0] > days:
        raise ValueError("%s has %d days; the hub draws its day from %d (renewable.py:51-53)" % (name, a.shape[0], days))
    if a.shape[0] < days:
        a = a[np.arange(days) % a.shape[0]]
    return np.ascontiguousarray(a, dtype="<f8")


def check_cdf(cdf):
    """arrival CDF table: [96, 301], every row non-decreasing within [0, 1] (CHS.hpp:731-743 scans it for the first >= u)"""
    c = np.asarray(cdf, dtype=np.float64)
    if c.shape != (SLOTS, CDF_COLS):
        raise ValueError("arrival CDF must have shape (%d, %d), got %s" % (SLOTS, CDF_COLS, c.shape))
    if c.min() < 0 or c.max() > 1.0 + 1e-6:
        raise ValueError("arrival CDF values must lie in [0, 1]")
    if np.any(np.diff(c, axis=1) < -1e-7):
        raise ValueError("arrival CDF rows must be non-decreasing")
    return c


def cdf_from_rates(rates, thin=1.0):
    """Poisson arrival CDFs from 96 mean vehicle counts per slot (the table the reference ships is of this kind):
    cdf[t][j] = P(Poisson(rates[t] * thin) <= j), j = 0..300."""
    lam = np.asarray(rates, dtype=np.float64).reshape(SLOTS) * float(thin)
    if np.any(lam < 0):
        raise ValueError("rates must be non-negative")
    j = np.arange(CDF_COLS, dtype=np.float64)
    from math import lgamma

    lg = np.array([lgamma(x + 1.0) for x in j])
    with np.errstate(divide="ignore"):
        logp = -lam[:, None] + j[None, :] * np.log(np.maximum(lam[:, None], 1e-300)) - lg[None, :]
    p = np.exp(logp)
    p[lam == 0, 0] = 1.0
    return np.minimum(np.cumsum(p, axis=1), 1.0)


def write_cdf_csv(path, cdf):
    """one row per slot of day, 301 comma-separated decimals -- the layout Read2Vector parses (CHS.hpp:96-155).  The
    runtime reads the text with the reference's own float parser, so what counts is the decimal string written here."""
    c = check_cdf(cdf)
    with open(path, "w") as f:
        for row in c:
            f.write(",".join("%.8f" % min(max(v, 0.0), 1.0) for v in row))
            f.write("\n")


def write_data_dir(out_dir, arrival_cdf=None, price=None, pv=None, wd=None, base_dir=None):
    """Create a libchub data directory; returns its path.  ``arrival_cdf`` may also be a path to a CSV in the
    reference's layout, which is copied byte for byte."""
    base = base_dir or _lib.DATA_DIR
    os.makedirs(out_dir, exist_ok=True)
    for name in _STATIC:
        shutil.copyfile(os.path.join(base, name), os.path.join(out_dir, name))
    dst = os.path.join(out_dir, _CDF)
    if arrival_cdf is None:
        shutil.copyfile(os.path.join(base, _CDF), dst)
    elif isinstance(arrival_cdf, (str, os.PathLike)) and os.fspath(arrival_cdf).lower().endswith(".csv"):
        shutil.copyfile(os.fspath(arrival_cdf), dst)
    else:
        write_cdf_csv(dst, load_series(arrival_cdf))
    if price is None:
        shutil.copyfile(os.path.join(base, "price_96.f64"), os.path.join(out_dir, "price_96.f64"))
    else:
        p = load_series(price).reshape(-1)
        if p.shape != (SLOTS,):
            raise ValueError("price must have %d values, got %s" % (SLOTS, p.shape))
        if np.std(p) == 0:
            raise ValueError("price series must not be constant (the observation divides by its std, MGR:45-46)")
        p.astype("<f8").tofile(os.path.join(out_dir, "price_96.f64"))
    for name, src, days, label in (("pv_100x96.f64", pv, PV_DAYS, "pv"), ("wd_150x96.f64", wd, WD_DAYS, "wd")):
        if src is None:
            shutil.copyfile(os.path.join(base, name), os.path.join(out_dir, name))
        else:
            _days(load_series(src), days, label).tofile(os.path.join(out_dir, name))
    return out_dir


__all__ = ["load_series", "check_cdf", "cdf_from_rates", "write_cdf_csv", "write_data_dir"]
