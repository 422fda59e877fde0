"""bench.py itself on the GPU box, in the short form the driver uses (--steps 20 --warmup 5): the one JSON line and what it must
carry; and the N > 1 code path (communicator, per-step gather, steps issued from C) on a world of one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"] + list(flags), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE JSON line
    return json.loads(lines[0])


def test_bench_short_form_prints_the_contract_line():
    d = _bench("--no-c5", "--no-cpu-baseline")
    assert d["metric"].startswith("env-steps/sec") and d["unit"] == "env-steps/s" and d["n_gpus"] == 1
    assert d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "65536 envs x hub [20 fast, 25 slow]" in d["config"]["workload"]
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["value"] > 5e8                      # an MI355X steps this workload at > 2 G env-steps/s; a tenth of that = broken
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.2 < r["frac"] < 1.0 and r["launches_sampled"] == 96
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the layout's own compulsory bytes (12 B per slot + 48 B per env) next to SURVEY 8(d)'s, and what the counters say limits the kernel
    assert r["layout_bytes_per_launch"] == 65536 * (12 * 45 + 48) and r["limited_by"].startswith("latency/issue")
    assert abs(r["frac_of_layout"] - r["layout_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-12 and r["frac_of_layout"] < r["frac"]
    # the day average does not depend on where the 20 timed steps fell: the same number the long default run reports (+- box noise)
    assert 15.0 < r["avg_launch_us"] < 30.0
    assert d["build_id"] and d["config"]["kernels_per_step"].startswith("2")
    # the three readings of the dominant kernel side by side (traffic: null unless profiles/ holds counters of this very build)
    assert "frac_of_traffic" in r and (r["frac_of_traffic"] is None) == (r["traffic"] is None)
    if r["traffic"] is not None:
        assert abs(r["frac_of_traffic"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-12 and r["frac_of_traffic"] < r["frac"]
    # ... and the rate over ten whole days next to the night-window `value` of the 20 timed steps
    assert d["day_avg"]["days"] == 10 and d["day_avg"]["launch"].startswith("hipGraph replays")
    assert abs(d["value_day_avg"] - 65536 / (d["ms_per_step_day_avg"] * 1e-3)) < 1e-6 * d["value_day_avg"]
    assert 0.8 * d["value"] < d["value_day_avg"] < 1.1 * d["value"] and d["value_day_avg"] > 5e8
    # ... and about two seconds of the same graph replays: GPU time an outside clock can see, at the ten days' rate
    su = d["sustained"]
    assert 1.0 < su["seconds"] < 5.0 and su["steps"] % 192 == 0 and abs(su["value"] - 65536 * su["steps"] / su["seconds"]) < 1e-6 * su["value"]
    assert abs(su["value"] / d["value_day_avg"] - 1) < 0.05


def test_bench_compat_block_carries_the_end_state_of_the_reference_exact_run():
    """`roofline_compat` (the reference-exact COMPAT mode at 65 536 envs: k_slot_walk2 + k_env) carries a digest of what its days left behind --
    observations, rewards, slot state, station records, the streams; the same days on the one-kernel-per-station form (the unit's first lane
    walks the env's two streams in the reference's consumption order, CHS.hpp:1272-1316 / 1583-1627; pinned to the reference's fixtures by
    test_compat_matches_reference_golden[*-4_per_station]) must leave the same digest: the rate the line prints is the rate of a correct run."""
    d = _bench("--no-cpu-baseline", "--no-bits")
    rc = d["roofline_compat"]
    assert rc["days_stepped"] == 8 and len(rc["end_state_digest"]) == 32 and 0.1 < rc["frac"] < 1.0
    assert "traffic" in rc and (rc["traffic"] is None or rc["traffic_over_algorithmic"] > 0.3)
    sys.path.insert(0, ROOT)
    import bench
    import charginghub_env_amd as chub
    from charginghub_env_amd import multi_gpu
    ref = bench.CompatBatch(chub, multi_gpu, 65536, 0, slot_kernel="wave")
    ref.days(rc["days_stepped"])
    want = ref.end_state_digest()
    ref.close()
    assert rc["end_state_digest"] == want


def test_bench_multi_rank_path_on_a_world_of_one():
    d = _bench("--force-comm", "--graph", "off", "--no-c5", "--no-cpu-baseline", "--no-events")
    assert d["n_ranks_seen"] == 1 and d["rccl_comm_count"] == 1 and len(d["ranks"]) == 1 and d["ranks"][0]["rank"] == 0
    assert "ncclSend/ncclRecv" in d["config"]["collective"] and d["value"] > 2e8
    assert d["roofline"] is None                  # --no-events: no per-kernel block, never a made-up one
    assert d["config"]["graph"].startswith("off (")
    ph = d["phases"]                              # the step taken apart per rank + what the builder expects of it
    assert len(ph["per_rank"]) == 1 and ph["per_rank"][0]["rank"] == 0 and 0 <= ph["per_rank"][0]["gather_us"] < 50  # (a world of one: the root's block is in place, nothing moves)
    assert ph["per_rank"][0]["host_issue_us"] > 0 and ph["expected_ms_per_step"] > 0 and ph["measured_ms_per_step"] == d["ms_per_step"]


def test_bench_multi_rank_path_reports_its_phases_with_kernel_times():
    d = _bench("--force-comm", "--graph", "off", "--no-c5", "--no-cpu-baseline", "--envs", "8192")
    r0 = d["phases"]["per_rank"][0]               # an 8-GPU shard's worth of envs: the single-launch step, priced as one kernel
    assert 3.0 < r0["slot_kernel_us"] + r0["env_kernel_us"] < 40.0 and 0.0 <= r0["gather_us"] < 200.0
    assert d["phases"]["expected_bound"] in ("gpu (kernels + gather)", "host issue (call by call)")
    # the expectation is built from the parts and the measurement must not be far under it (nothing runs faster than its parts)
    assert d["phases"]["measured_ms_per_step"] > 0.5 * d["phases"]["expected_ms_per_step"]


def test_bench_overlapped_gather_in_a_graph_on_a_world_of_one():
    """--graph on --overlap-gather: the N > 1 graph form with the gather on the communicator's own stream (graph edges); the line says so and
    its expectation is the larger of kernels and gather, not their sum"""
    d = _bench("--force-comm", "--graph", "on", "--overlap-gather", "--no-c5", "--no-cpu-baseline", "--envs", "8192", "--steps", "192", "--warmup", "96")
    assert "communicator's own stream" in d["config"]["collective"] and d["config"]["graph"] == "on"
    r0, ph = d["phases"]["per_rank"][0], d["phases"]
    assert ph["expected_bound"].startswith("gpu (the larger of kernels and gather")
    assert abs(ph["expected_ms_per_step"] * 1e3 - max(r0["slot_kernel_us"] + r0["env_kernel_us"], r0["gather_us"])) < 1e-6
    s = _bench("--force-comm", "--graph", "on", "--no-c5", "--no-cpu-baseline", "--envs", "8192", "--steps", "192", "--warmup", "96")
    assert d["value"] > 0.7 * s["value"]          # measured 0.95 x the serial graph form on one GPU (the "gather" is a local copy that competes with the kernels)
