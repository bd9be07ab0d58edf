"""bench.py's bookkeeping that needs no GPU: the metric's name per launch shape, the plausibility bound on a shard's survivors, and
the error line a failed run leaves (`python bench.py --gpus N` always ends in ONE JSON line: tests/test_bench_rehearsal_gpu.py
checks that on the device; here the pieces)."""
import argparse
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rv_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _args(**kw):
    base = dict(gpus=1, steps=20, warmup=3, rows=bench.ROWS_PER_GPU, scaling="weak", global_rows=1e10, workload="filter_project")
    base.update(kw)
    return argparse.Namespace(**base)


def test_metric_names_the_launch_shape():
    assert bench.metric_name(_args(), 1) == "rows/sec filter+project, 1e9-row Int64, 10% selectivity"   # BASELINE.json's metric, verbatim
    weak8 = bench.metric_name(_args(gpus=8), 8)
    assert "PER GPU x 8 GPUs (weak scaling)" in weak8 and weak8.startswith("rows/sec filter+project")
    assert "not the headline" in bench.metric_name(_args(scaling="strong"), 8)
    assert "not the headline" in bench.metric_name(_args(workload="filter_agg"), 1)


def test_survivor_plausibility_is_six_sigma_around_the_generators_rate():
    n = 1_000_000_000
    assert bench.survivors_plausible("filter_project", n, 99_989_506)          # the headline's exact count
    assert not bench.survivors_plausible("filter_project", n, 0)               # a rank that filtered nothing
    assert not bench.survivors_plausible("filter_project", n, 2 * 99_989_506)  # ... or its shard twice
    assert not bench.survivors_plausible("filter_project", n, 100_100_000)     # 10 sigma off
    assert bench.survivors_plausible("and2_nulls", n, 90_238_938)
    assert bench.survivors_plausible("filter_project", 0, 0) and not bench.survivors_plausible("filter_project", 0, 1)


def test_error_line_carries_phase_and_per_rank_state(capsys):
    prog = bench.Progress(_args(gpus=4))
    prog.ranks[0] = {"rank": 0, "device": 0, "rows": 10}
    prog.ranks[2] = {"rank": 2, "device": 2, "rows": 10, "survivors": 1}
    prog.at("timed")
    line = prog.error_line("rank 2: RvError: RV_ERR_OOM")
    assert line["phase"] == "timed" and line["value"] is None and line["n_gpus"] == 4 and line["unit"] == "rows/s"
    assert line["per_rank"][1] is None and line["per_rank"][2]["survivors"] == 1   # ranks that reported nothing stay visible as gaps
    json.dumps(line)  # serialisable as it stands


def test_fault_injection_spec_names_a_rank_and_a_phase(monkeypatch):
    monkeypatch.setenv("RV_BENCH_FAIL", "3:warmup")
    prog = bench.Progress(_args(gpus=4))
    assert prog.at("generate") is None and prog.at("warmup") == 3 and prog.phase == "warmup"


def test_a_run_that_cannot_start_still_prints_one_line():
    """No GPU here: the single-process driver fails while creating the group -- and says so in one JSON line, exit code 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "1000"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, r.stdout + r.stderr[-1500:]
    line = json.loads(lines[0])
    assert line["value"] is None and line["phase"] == "group_create" and line["n_gpus"] == 2 and "error" in line
