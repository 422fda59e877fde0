#!/usr/bin/env python3
"""Random hub shapes with stations of more than 256 piles (k_slot_unit_any) through the parity checks of tests/test_gpu_big_stations.py:
PHILOX and COMPAT handles against liboracle_big.so, bit for bit.  usage (GPU box, repo root): python3 tools/experiments/big_shape_sweep.py --seed 1 --shapes 12"""
import argparse, os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--shapes", type=int, default=12)
args = ap.parse_args()
import orclib
import test_gpu_parity as parity
import test_gpu_big_stations as big
rs = np.random.RandomState(args.seed)
for i in range(args.shapes):
    a = int(rs.choice([rs.randint(257, 1400), rs.randint(257, 520), 256 * rs.randint(2, 6) + rs.randint(-1, 2)]))
    b = int(rs.choice([rs.randint(0, 65), rs.randint(65, 257), rs.randint(257, 900)]))
    piles = [a, b] if rs.randint(2) else [b, a]
    types = [["fast", "slow"], ["slow", "fast"], ["fast", "fast"], ["slow", "slow"]][rs.randint(4)]
    n = int(rs.randint(1, 5))
    kw = dict(big.BIG_KW, station_list=piles, station_type_list=types, constant_charging=bool(rs.randint(4) == 0))
    with orclib.big_oracle(parity):
        parity._philox_parity("sweep_%d" % i, kw, n, plan=(30, 12))                            # the packed kernel where the hub fits a tile
        parity._philox_parity("sweep_%d_wave" % i, kw, n, plan=(30,), slot_kernel="wave")      # ... and the unit / chunked kernels
    big.compat_parity(piles, types, n, 40, cc=kw["constant_charging"], seed=100 + i)
    print("shape", i, piles, types, "envs", n, "cc", kw["constant_charging"], "ok", flush=True)
print("all", args.shapes, "shapes equal")
