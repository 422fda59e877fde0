"""CPU-side checks of the measurement code (bench.py): the accounting it prices the kernels with, the rule
by which a committed PMC profile is accepted as this build's traffic, and the shape of the committed bench lines."""
import glob
import importlib.util
import json
import os
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("chub_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_are_the_survey_formula():
    b = _bench()
    for S, D, total in ((16, 9, 881), (45, 13, 1941), (64, 13, 2625)):  # C2, C3 / C4, C5
        slot, env = b.algorithmic_bytes(S, D)
        assert slot + env == 36 * S + 269 + 4 * D == total
        assert slot == 36 * S + 34
    assert b.HBM_PEAK_GBS == 8000.0
    assert b.CONFIGS["c4"][0] == 65536 and b.CONFIGS["c5"][0] == 262144 and b.CONFIGS["c5"][1]["station_list"] == [32, 32]


def test_traffic_profiles_are_matched_by_build_and_workload():
    """a PMC profile counts as this build's traffic only with the same build id (sources + kernel knobs) and workload"""
    b = _bench()
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")))  # the last one: the latest round's final set
    assert profs and os.path.basename(profs[-1]).startswith("round6_"), "no round-6 traffic profile committed"
    rec = json.load(open(profs[-1]))
    got, src = b.measured_traffic(rec["build_id"], 65536, 65536, [20, 25])
    assert src == os.path.basename(profs[-1]) and abs(got - rec["k_slot"]["traffic_bytes_per_launch"]) < 1.0
    half, _ = b.measured_traffic(rec["build_id"], 32768, 65536, [20, 25])      # a shard's share of the same job
    assert abs(half - got / 2) < 1.0
    assert b.measured_traffic("not-a-build", 65536, 65536, [20, 25]) == (None, None)
    assert b.measured_traffic(rec["build_id"], 65536, 65536, [32, 32]) == (None, None)  # another hub is another workload
    c5 = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic_c5.json")))
    if c5:
        r5 = json.load(open(c5[-1]))
        assert b.measured_traffic(r5["build_id"], 262144, 262144, [32, 32])[1] == os.path.basename(c5[-1])
    # the profile is only live while it is the profile of the sources in the tree: say so when it is not (no failure: the next
    # kernel change makes it stale until tools/refresh_profiles.sh has run on a GPU box)
    import sys
    sys.path.insert(0, ROOT)
    from charginghub_env_amd import _lib
    if rec["build_id"] != _lib.source_hash():
        warnings.warn("profiles/%s was measured on build %s, the sources are %s: bench.py will report roofline.traffic = null "
                      "until the profile set is refreshed" % (os.path.basename(profs[-1]), rec["build_id"], _lib.source_hash()))


def test_committed_bench_line_has_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench.json")))
    assert lines and os.path.basename(lines[-1]).startswith("round6_")
    d = json.load(open(lines[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "n_ranks_seen"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["launches_sampled"] == 96 and "each slot of the day once" in r["window"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "env-steps/s" and c["sample"]
    assert abs(d["value"] - 65536 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert "roofline_c5" in d and d["roofline_c5"]["state_bytes"] == 262144 * 64 * 4  # one 32-bit word per slot since round 4
    assert d["roofline"]["traffic"] and d["roofline"]["traffic_source"] == os.path.basename(lines[-1]).replace("_bench.json", "_pmc_traffic.json")
    assert d["dropin_single_env"]["compat_us_per_step"] < d["dropin_single_env"]["reference_us_per_step"]
    # round 5: the layout's own compulsory bytes next to SURVEY 8(d)'s, what the counters say limits the kernel, the COMPAT step priced
    assert r["layout_bytes_per_launch"] == 65536 * (12 * 45 + 48) and r["limited_by"].startswith("latency/issue")
    assert abs(r["frac_of_layout"] - r["layout_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-12
    assert d["roofline_c5"]["limited_by"] == "hbm" and d["roofline_c5"]["layout_bytes_per_launch"] == 262144 * (12 * 64 + 48)
    rc = d["roofline_compat"]
    assert rc["bound"] == "hbm" and abs(rc["frac"] - 1941 * 65536 / ((rc["slot_pass_and_next_walks_us"] + rc["tails_us"]) * 1e-6) / 8e12) < 1e-9
    # round 6: the three readings of the dominant kernel side by side, the rate over ten whole days, COMPAT's counter traffic and end-state digest
    assert abs(r["frac_of_traffic"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-12 and r["frac_of_layout"] < r["frac_of_traffic"] < r["frac"]
    assert d["day_avg"]["days"] == 10 and abs(d["value_day_avg"] - 65536 / (d["ms_per_step_day_avg"] * 1e-3)) < 1e-6 * d["value_day_avg"]
    assert abs(d["value_day_avg"] / d["value"] - 1) < 0.05
    assert d["sustained"]["seconds"] > 1.0 and abs(d["sustained"]["value"] / d["value_day_avg"] - 1) < 0.03
    assert rc["traffic_source"] == os.path.basename(lines[-1]).replace("_bench.json", "_pmc_compat.json") and 1.0 < rc["traffic_over_algorithmic"] < 2.5
    assert len(rc["end_state_digest"]) == 32 and rc["days_stepped"] == 8
    assert "COMPAT" in d["cpu_baseline"]["sample"]
    assert d["cpu_baseline"]["host_cores"] >= d["cpu_baseline"]["cores"]
    # ... and the rocprofv3 summaries the three fractions can be recomputed from are committed next to the line
    for name in ("_kernel_stats.csv", "_kernel_stats_c5.csv", "_kernel_stats_compat.csv", "_pmc_compat.json", "_pmc_traffic_c5.json"):
        assert os.path.exists(lines[-1].replace("_bench.json", name)), name


def test_reference_stations_rate_is_timed_where_the_reference_build_exists():
    """cpu_baseline.reference_stations: the unmodified CHS.hpp (oracle/_ref) timed on one core beside the port -- present wherever the
    library travelled, a plausible rate (the survey measured 12.8 k hub-steps/s for this hub on one core of this container)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    b = _bench()
    r = b.reference_stations_rate(dict(station_list=[20, 25], station_type_list=["fast", "slow"], fcev_permeate=0.01), target_seconds=0.5)
    if not orclib.ref_available():
        assert r is None
        return
    assert r["kind"] == "reference" and r["cores"] == 1 and r["unit"] == "hub-steps/s" and 1e3 < r["value"] < 1e6 and "CHS.hpp" in r["sample"]
    # a hub with a station without piles times the other station alone
    r0 = b.reference_stations_rate(dict(station_list=[16, 0], station_type_list=["fast", "slow"], fcev_permeate=0.0), target_seconds=0.3)
    assert r0["value"] > r["value"]
