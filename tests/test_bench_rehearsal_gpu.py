"""bench.py's N > 1 protocol end to end on ONE GPU (-m gpu): two ranks launched with torch.distributed.run, both on
device 0 (RV_BENCH_ONE_DEVICE=1: gloo instead of RCCL, which refuses two ranks on one device).  Exercises what the
driver's 8-GPU run exercises -- row-range shards of ONE global table (--scaling strong, BASELINE configs[3]), the
per-rank fused pass, the survivor-count exchange, the rank-order gather into one shared pinned host buffer, and the
{SUM, COUNT} reduction of configs[4] -- and checks the line against the oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra, ranks=2, fail=None):
    env = dict(os.environ, RV_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fail:
        env["RV_BENCH_FAIL"] = fail
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert (r.returncode != 0) == bool(fail), r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    return json.loads(lines[0])


def _run_single_process(extra, ranks=2, fail=None):
    """`python3 bench.py --gpus N ...` with NO launcher (RANK / WORLD_SIZE unset): the single-process rv_group driver."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RV_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fail:
        env["RV_BENCH_FAIL"] = fail
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert (r.returncode != 0) == bool(fail), r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("phase", ["generate", "warmup", "timed"])
def test_a_failing_rank_leaves_an_error_line_not_a_traceback(phase):
    """The first contact with an 8-GPU node must leave a record whatever happens: a rank that fails (out of memory on one device,
    a device fault -- here injected) ends the run with ONE JSON line {"error", "phase", per-rank state} and a non-zero exit
    code, under both launch protocols, and nobody is left waiting in a collective."""
    line = _run_single_process(["--rows", "6000000"], ranks=3, fail=f"1:{phase}")
    order = ["generate", "warmup", "timed"]  # (the injected fault fires in the rank's next QUERY call: armed before `generate`, it hits the warm-up)
    assert line["value"] is None and line["n_gpus"] == 3 and order.index(line["phase"]) in (order.index(phase), order.index(phase) + 1), line
    assert "injected" in line["error"] or "RV_ERR_DEVICE" in line["error"], line["error"]
    assert len(line["per_rank"]) == 3 and all(r["device"] == 0 for r in line["per_rank"])
    if phase != "generate":
        assert all(r["rows"] == 6_000_000 and r["hbm_free_bytes"] > 0 for r in line["per_rank"]), line["per_rank"]
    line = _run(["--rows", "4000000"], fail=f"1:{phase}")
    assert line["value"] is None and line["n_gpus"] == 2 and line["phase"] == phase and "rank 1" in line["error"]
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["rows"] == 4_000_000 for r in line["per_rank"])


def test_gpus_n_without_a_launcher_runs_the_single_process_driver(oracle):
    """The driver's N = 1 command is plain `python3 bench.py --gpus 1 ...`; the same spelling with N > 1 must produce a line,
    not an error: one process, rv_group over the N devices (here: device 0 listed twice)."""
    line = _run_single_process(["--rows", "6000000"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_rows"] == 12_000_000
    assert "PER GPU x 2 GPUs (weak scaling)" in line["metric"]  # N > 1: the weak-scaling reading is spelled out ...
    s1 = line["strong_1e9"]                                      # ... next to the strong-scaling one (ONE table of --rows rows)
    assert s1["global_rows"] == 6_000_000 and s1["rows_per_gpu"] == 3_000_000 and s1["value"] > 0
    assert "single process" in line["driver"] and line["rccl_ranks"] == 0
    pr = line["per_rank"]
    assert [r["rows"] for r in pr] == [6_000_000, 6_000_000] and all(r["kernel_ms"] > 0 and r["survivors_plausible"] for r in pr)
    want = oracle.eval_predicate([oracle.generate(synth_spec(RV_INT64, seed=42, length=12_000_000))], Predicate([Term(0, ">", 899)]))[1]
    assert round(line["config"]["selectivity"] * 12_000_000) == want
    assert abs(line["value"] - 12_000_000 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["roofline"]["kernel_id"] == "fused_filter_compact<1,16,2,16,32>" and line["roofline"]["kernel_ms_avg"] > 0
    e2e = line["end_to_end"]
    assert "error" not in e2e and e2e["gathered_bytes_per_step"] == want * 8 and e2e["note"].endswith("ok")
    assert e2e["filter_ms"] > 0 and e2e["gather_ms"] > 0
    assert sum(r["survivors"] for r in pr) == want
    assert "see" in line["cpu_baseline"] and "value" not in line["cpu_baseline"]  # timed on rank 0 at N = 1 only: a pointer here


def test_single_process_strong_scaling_and_aggregate(oracle):
    n = 20_000_037
    line = _run_single_process(["--scaling", "strong", "--global-rows", str(n)], ranks=3)
    want = oracle.eval_predicate([oracle.generate(synth_spec(RV_INT64, seed=42, length=n))], Predicate([Term(0, ">", 899)]))[1]
    assert line["n_gpus"] == 3 and line["config"]["global_rows"] == n and round(line["config"]["selectivity"] * n) == want
    assert line["end_to_end"]["note"].endswith("ok")
    agg = _run_single_process(["--scaling", "strong", "--global-rows", str(n), "--workload", "filter_agg"])
    assert round(agg["config"]["selectivity"] * n) == want and agg["end_to_end"] is None
    assert agg["roofline"]["kernel_id"].startswith("filter_agg_kernel<1,16,2,4")
    assert agg["allreduce"]["path"].startswith("host sum") and agg["rccl_ranks"] == 0  # one device listed twice: no communicator
    a3 = _run_single_process(["--rows", "3000000", "--workload", "and2_nulls"])
    assert a3["dtype"] == "f64+int64" and 0.085 < a3["config"]["selectivity"] < 0.095 and "error" not in a3["end_to_end"]


def test_n1_line_carries_the_launched_kernel_and_matching_traffic():
    line = _run(["--rows", "5000000"], ranks=1)
    assert line["roofline"]["kernel_id"] == "fused_filter_compact<1,16,2,16,32>"
    assert line["roofline"]["traffic"] is None  # not the profiled size: no constant attached


def test_strong_scaling_two_ranks_gather_in_rank_order(oracle):
    n = 20_000_037
    line = _run(["--scaling", "strong", "--global-rows", str(n)])
    want = oracle.eval_predicate([oracle.generate(synth_spec(RV_INT64, seed=42, length=n))], Predicate([Term(0, ">", 899)]))[1]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["global_rows"] == n
    assert round(line["config"]["selectivity"] * n) == want
    e2e = line["end_to_end"]
    assert "error" not in e2e and e2e["gathered_bytes_per_step"] == want * 8 and e2e["note"].endswith("ok")
    assert line["kernel_only"]["value"] == line["value"] and e2e["value"] <= line["value"] * 1.05
    assert line["roofline"]["kernel"].startswith("fused_filter_compact<1,16,2,16")


def test_filter_agg_two_ranks_reduce_to_the_global_count(oracle):
    n = 10_000_019
    line = _run(["--scaling", "strong", "--global-rows", str(n), "--workload", "filter_agg"])
    want = oracle.filter_agg([oracle.generate(synth_spec(RV_INT64, seed=42, length=n))], Predicate([Term(0, ">", 899)]), 0)[2]
    assert round(line["config"]["selectivity"] * n) == want
    assert line["dtype"] == "int64" and "configs[4]" in line["config"]["workload"] and line["end_to_end"] is None
    assert line["roofline"]["kernel"].startswith("filter_agg_kernel")


def test_config2_workload_line_describes_itself():
    line = _run(["--workload", "and2_nulls", "--rows", "3000000"], ranks=1)
    assert line["dtype"] == "f64+int64" and "configs[2]" in line["config"]["workload"]
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 16.25 * 3_000_000
    assert 0.085 < line["config"]["selectivity"] < 0.095 and "error" not in line["end_to_end"]


def test_weak_scaling_two_ranks_line_as_the_driver_launches_it():
    """`bench.py --gpus 2 --steps K --warmup W` under torch.distributed.run: whole-job value, weak scaling, no end-to-end leg
    unless asked for (its collectives would leave the other ranks waiting if one failed inside it)."""
    line = _run(["--rows", "4000000"])
    assert line["scaling"] == "weak" and line["n_gpus"] == 2 and line["config"]["global_rows"] == 8_000_000
    assert "PER GPU x 2 GPUs (weak scaling)" in line["metric"] and line["strong_1e9"]["global_rows"] == 4_000_000
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["survivors_plausible"] for r in line["per_rank"])
    assert round(line["config"]["selectivity"] * 8_000_000) == sum(r["survivors"] for r in line["per_rank"])
    assert line["end_to_end"] is None and line["kernel_only"]["value"] == line["value"]
    assert abs(line["value"] - 8_000_000 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    forced = _run(["--rows", "4000000", "--end-to-end"])
    assert "error" not in forced["end_to_end"] and forced["end_to_end"]["note"].endswith("ok")


def test_host_table_workload_three_ranks_on_one_device():
    """bench.py --workload host_table: rv_group_filter_project_host -- a pinned host table cut into row ranges, every range through its
    rank's own chunk pipeline, survivors gathered in rank order (unmeasured on hardware for N > 1: the ranks share ONE link here)."""
    line = _run_single_process(["--workload", "host_table", "--rows", "3000000"], ranks=3)
    assert line["n_gpus"] == 3 and line["config"]["global_rows"] == 9_000_000 and line["check"].endswith("ok")
    assert len(line["pcie"]["per_rank_upload_gb_s"]) == 3 and all(g > 0 for g in line["pcie"]["per_rank_upload_gb_s"])
    assert abs(line["config"]["selectivity"] - 0.1) < 0.002
    # a rank that fails leaves the error line
    line = _run_single_process(["--workload", "host_table", "--rows", "3000000"], ranks=3, fail="2:timed")
    assert line["error"] and line["phase"] == "timed" and "injected failure" in line["error"]
