#!/usr/bin/env python3
"""Convert the reference's data assets into the flat files the runtime loads.

Run in the build container only (reads /root/reference, which does not exist on
the GPU box).  Outputs go to charginghub-env_amd/data/:

  car_flow_possibility_list_save.csv  arrival CDFs, 96 rows x 301 cols, copied
                                      byte-for-byte (LF flavour, test/ copy) --
                                      the runtime parses it with the reference's
                                      own non-standard float parser (CHS.hpp:138-155)
  price_96.f64      96 little-endian float64          (Aggregator_Simple.py:9-15)
  pv_100x96.f64     100x96 little-endian float64      (renewable.py:10-13)
  wd_150x96.f64     150x96 little-endian float64      (renewable.py:15-18)

The pickles hold plain Python lists of float/int, so float64 is lossless.
"""
import os
import pickle
import shutil
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "charginghub-env_amd", "data")


def main():
    os.makedirs(OUT, exist_ok=True)
    d = os.path.join(REF, "evcssp_env_cpp", "envs", "data_file")
    price = pickle.load(open(os.path.join(d, "price_data", "price_after_MAD_96.pkl"), "rb"))
    pv = pickle.load(open(os.path.join(d, "pv_power_100.pkl"), "rb"))
    wd = pickle.load(open(os.path.join(d, "wd_power_150.pkl"), "rb"))
    price = np.asarray(price, dtype="<f8")
    pv = np.asarray(pv, dtype="<f8")
    wd = np.asarray(wd, dtype="<f8")
    assert price.shape == (96,) and pv.shape == (100, 96) and wd.shape == (150, 96)
    price.tofile(os.path.join(OUT, "price_96.f64"))
    pv.tofile(os.path.join(OUT, "pv_100x96.f64"))
    wd.tofile(os.path.join(OUT, "wd_150x96.f64"))
    shutil.copyfile(os.path.join(REF, "test", "car_flow_possibility_list_save.csv"),
                    os.path.join(OUT, "car_flow_possibility_list_save.csv"))
    os.chmod(os.path.join(OUT, "car_flow_possibility_list_save.csv"), 0o644)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
