"""Stations of more than 256 piles on the GPU (round 6): the reference's station constructors take any pile count
(CHS.hpp:1148, 1458); libchub walks such a unit in chunks of 256 piles (k_slot_unit_any, one workgroup per (env, station), up to
4096 piles per station).  Held to
  (a) the reference itself: the fixture env_big_300_270 (300 fast + 270 slow piles, recorded from the unmodified reference by
      oracle/gen/gen_env_golden.py) through COMPAT handles, restore into a fresh handle included, and
  (b) the oracle on the same seeds -- liboracle_big.so, the same source with room for 4096 piles per station, itself pinned against the
      reference's stations of 300 ... 4096 piles (tests/test_oracle_vs_ref.py::*_beyond_256_piles) -- through the checks the smaller
      hubs go through: PHILOX steps and resets, the scalar-load control in both RNG modes, calls on subsets of the envs."""
import numpy as np
import pytest

import orclib
import test_gpu_env_clocks as clocks
import test_gpu_parity as parity

pytestmark = pytest.mark.gpu

BIG_KW = dict(hydro_prod_rate=2000.0, hydro_store_vlt=5000.0, init_soc=0.5, fc_max_power=100.0, fcev_permeate=0.02)


@pytest.mark.parametrize("name", orclib.GOLDEN_ENV_BIG)
@pytest.mark.parametrize("n_envs,migrate", [(1, False), (3, True), (5, False)], ids=["1", "3_restored", "5"])
def test_compat_matches_reference_golden_on_stations_of_more_than_256_piles(name, n_envs, migrate):
    parity.test_compat_matches_reference_golden(name, n_envs, migrate)


@pytest.mark.parametrize("label,piles,types,n,plan", [
    ("300_20", [300, 20], ["fast", "slow"], 5, (96, 30)),        # one chunked unit beside a wave-local one
    ("70_600", [70, 600], ["slow", "fast"], 3, (96, 20)),        # k_slot_unit beside k_slot_unit_any, three chunks
    ("257_0", [257, 0], ["slow", "fast"], 4, (40, 10)),          # one pile into the second chunk
    ("512_1000", [512, 1000], ["fast", "slow"], 2, (30, 10)),    # whole chunks; evs_reset admits ~ 500 cars (balk places beyond 690)
    ("4096_3", [4096, 3], ["slow", "fast"], 1, (12, 6)),         # the largest station the library takes
])
def test_philox_matches_oracle_on_stations_of_more_than_256_piles(label, piles, types, n, plan):
    kw = dict(BIG_KW, station_list=piles, station_type_list=types)
    with orclib.big_oracle(parity):
        parity._philox_parity("big_" + label, kw, n, plan=plan)


@pytest.mark.parametrize("cc", [False, True])
@pytest.mark.parametrize("rng", ["philox", "compat"])
@pytest.mark.parametrize("piles", [(300, 20), (3, 700)], ids=["300_20", "3_700"])
def test_scalar_load_mode_matches_oracle_on_stations_of_more_than_256_piles(piles, rng, cc):
    """evs_step(float): assign_on_off's urgency order (CHS.hpp:1318-1362 / 1629-1674) is a rank over the WHOLE unit"""
    with orclib.big_oracle(parity):
        parity.test_scalar_load_mode_matches_oracle(piles, rng, cc)


def test_subset_resets_and_steps_on_stations_of_more_than_256_piles(monkeypatch):
    """chub_reset_envs / chub_step_envs: the whole workgroup of a unit whose env is not named leaves it alone"""
    monkeypatch.setitem(clocks.SHAPES, "huge", (dict(clocks.KW, station_list=[300, 260]), "auto"))
    with orclib.big_oracle(parity, clocks):
        clocks.test_subset_resets_and_steps_match_the_oracle("huge")


def test_more_than_4096_piles_per_station_is_refused():
    chub = parity.hub()
    with pytest.raises(chub.ChubError) as ei:
        chub.VecChargingHub(1, rng="philox", station_list=[4097, 0], station_type_list=["fast", "slow"], **BIG_KW)
    assert "4096" in str(ei.value)
