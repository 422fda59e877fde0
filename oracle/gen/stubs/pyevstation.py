"""TEST INFRASTRUCTURE (fixture generation only).

Binding layer that gives the reference's Python files the module name they import (`pyevstation`,
Aggregator_Simple.py:7, hydro_sys.py:6, evcssp_manager.py:16).  In the reference that module is a
Boost.Python wrapper (lion_cpp20/main.cpp:19-290) whose Boost dependency is an un-vendored submodule;
here the same names forward, through ctypes, to the REAL reference C++ core compiled from its own
header (oracle/_ref/libchs_ref.so <- SCP_Base/CHS.hpp).  No arithmetic lives in this file.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))
import orclib  # noqa: E402

_r = orclib.ref()


def Change_Use_Seed(flag):
    _r.ref_change_use_seed(int(bool(flag)))


class Vector_float(list):
    pass


class _Station(object):
    _typ = None

    def __init__(self, piles, wait=True, constant_charging=False):
        self._n = piles
        self._h = _r.ref_station_new(self._typ, int(piles), int(wait), int(constant_charging))
        self._tl_override = None

    def _sc(self):
        out = np.zeros(8)
        _r.ref_station_scalars(self._h, orclib.ptr(out))
        return out

    charge_number = property(lambda self: self._n)
    min_power = property(lambda self: float(self._sc()[0]))
    charge_power = property(lambda self: float(self._sc()[1]))
    max_power = property(lambda self: float(self._sc()[2]))
    car_number = property(lambda self: int(self._sc()[3]))
    line = property(lambda self: int(self._sc()[4]))
    flow_in_number = property(lambda self: [int(self._sc()[5])])
    transformer_limit = property(lambda self: float(self._sc()[7]))

    def evs_step(self, actions):
        a = np.asarray(list(actions), dtype=np.float32)
        _r.ref_station_step(self._h, orclib.ptr(a), len(a))

    def evs_reset(self):
        _r.ref_station_reset(self._h)

    def slots(self):
        out = np.zeros((9, self._n), dtype=np.float32)
        _r.ref_station_slots(self._h, orclib.ptr(out))
        return out

    def print_situation(self):
        pass


class FastChargeStation(_Station):
    _typ = 0


class SlowChargeStation(_Station):
    _typ = 1


class PoissonNumber(object):
    @staticmethod
    def hv_car_number_wrt_poisson(time, possible_in, permeability):
        return _r.ref_hv(int(time), float(possible_in), float(permeability))


class CarArriveRandom(object):
    @staticmethod
    def mk_soc():
        return float(_r.ref_mk_soc())


class RandomUtil(object):
    @staticmethod
    def uniform_rand(a, b):
        return float(_r.ref_uniform_rand(float(a), float(b)))
