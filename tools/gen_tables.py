#!/usr/bin/env python3
"""Generate the two sampling tables of the PHILOX (production) RNG mode -> charginghub-env_amd/data/.

soc_d_icdf_4097.f32   Td[i] = 7 + 3*Phi^-1(i/4096), i = 1..4095 (Td[0], Td[4096]: the quantiles at 2^-22 from either end),
                      float32, NOT clipped.  The arrival "driver experience" of CarArriveRandom::mk_soc (CHS.hpp:804-814)
                      is N(7,3) clipped to [1,10]; PHILOX mode samples the normal by linear interpolation of this inverse CDF
                      with a 32-bit uniform (12 bits select the cell, 20 bits interpolate) and clips AFTERWARDS, as the
                      reference does -- so the two clip atoms have the law's mass to interpolation accuracy (clipped nodes
                      put each clip point off by up to one table cell, 2.4e-4).
late_thr_16.u32       LT[j] = floor(2^32 * P(max(0, round(N(2,2))) <= j)), j = 0..15, uint32.  mk_late_time
                      (CHS.hpp:816-830, the "slow" law both pile classes use) is then #{j : w >= LT[j]} for a
                      32-bit uniform w.

normal_icdf_4097.f32  Tz[i] = Phi^-1(i/4096), i = 1..4095 (Tz[0], Tz[4096] = the tail table's end points), float32.
normal_tail_4097.f32  TL[j] = Phi^-1(j * 2^-20), j = 1..4096, TL[0] = Phi^-1(2^-22), float32: second level for the
                      lowest 16 cells of Tz, p < 2^-8 (the highest 16 use it mirrored): the quantile function bends most
                      there, and interpolating Tz alone left a Kolmogorov distance of 2e-5 to the normal law (now 1.7e-6,
                      tests/test_law_fidelity_cpu.py).  A standard normal from one 32-bit uniform w: cell = w >> 20; inside
                      cells 16..4079 interpolate Tz linearly with the low 20 bits; cells 0..15 interpolate TL with w >> 12
                      and the low 12 bits; cells 4080..4095 are -f(~w).
                      Used for the three OU noises (REN:71-76) and the initial occupancy (CHS.hpp:832-842).

Both the runtime (libchub) and the oracle load these files, so the two sides share one definition.
"""
import os

import numpy as np
from scipy.special import ndtr, ndtri

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "charginghub-env_amd", "data")


def main():
    p = np.arange(4097, dtype=np.float64) / 4096.0
    with np.errstate(divide="ignore"):
        z = ndtri(p)
    z[0], z[-1] = ndtri(2.0 ** -22), -ndtri(2.0 ** -22)
    d = (7.0 + 3.0 * z).astype("<f4")
    assert d[0] < 1.0 and d[-1] > 10.0 and np.all(np.diff(d) > 0)
    d.tofile(os.path.join(OUT, "soc_d_icdf_4097.f32"))
    j = np.arange(16, dtype=np.float64)
    cdf = ndtr((j + 0.5 - 2.0) / 2.0)
    thr = np.minimum(np.floor(cdf * 4294967296.0), 4294967295.0).astype("<u4")
    assert np.all(np.diff(thr.astype(np.int64)) >= 0)
    thr.tofile(os.path.join(OUT, "late_thr_16.u32"))
    tail = np.arange(4097, dtype=np.float64) * 2.0 ** -20
    tail[0] = 2.0 ** -22
    tl = ndtri(tail).astype("<f4")
    tz = np.empty(4097, dtype=np.float64)
    tz[1:4096] = ndtri(p[1:4096])
    tz[0] = float(tl[0])
    tz[4096] = -float(tl[0])
    tz = tz.astype("<f4")
    assert tl[4096] == tz[16] and np.all(np.diff(tz) > 0) and np.all(np.diff(tl) > 0)
    tz.tofile(os.path.join(OUT, "normal_icdf_4097.f32"))
    tl.tofile(os.path.join(OUT, "normal_tail_4097.f32"))
    print("Tz[:3]", tz[:3], "TL[:3]", tl[:3], "TL[-1]", tl[-1])
    print("Td[:4]", d[:4], "Td[-4:]", d[-4:], "LT", thr)


if __name__ == "__main__":
    main()
