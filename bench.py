#!/usr/bin/env python3
"""bench.py -- rows/sec of filter+project over a 1e9-row Int64 column at 10 % selectivity
(BASELINE.json metric, configs[1]) on N MI355X GPUs of one node.

A step is ONE pass of the hot path over one resident batch: rv_filter_project(x > 899,
project [x]) over the rank's synthetic rows (already in HBM), i.e. the per-launch
ticket/descriptor memset, the fused single-pass kernel and the 512-byte result readback
that tells the host how many rows survived.  Ranks hold disjoint row ranges of one global
column (row-range shards, no data-path collective).

Three ways to run it, one JSON line each:

    python bench.py --gpus 1 --steps 20 --warmup 3          # one context; the headline (configs[1])
    python bench.py --gpus 8 --steps 20 --warmup 3          # NO launcher: ONE process drives the 8 GPUs through
                                                            # rv_group_* (one context + one host thread per device; the
                                                            # caller the north star names is a single Rust process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3     # one process per GPU (RANK / WORLD_SIZE set)

Weak scaling by default (1e9 rows per GPU).  Other BASELINE configs (never the headline `metric`; the line says
which workload ran):
    --workload and2_nulls                      configs[2]: (f > 0.5) AND (x < 200) over nullable Float64 + Int64
    --scaling strong [--global-rows 1e10]      configs[3]: ONE 1e10-row table cut into N row ranges; the line carries
                                               kernel-only (`value`) and `end_to_end` (+ rank-order gather of the
                                               survivors into one pinned host buffer)
    --workload filter_agg [--scaling strong]   configs[4]: filter + SUM/COUNT, RCCL all-reduce of 16 bytes
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_PER_GPU = 1_000_000_000
GLOBAL_ROWS_STRONG = 10_000_000_000
SEED_X = 42
LITERAL = 899  # x in [0, 1000): x > 899 keeps 10 %
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)

WORKLOADS = {
    # name: (description, dtype, read bytes/row, written bytes per surviving row, what the dominant kernel does)
    "filter_project": ("filter(x > 899).select([x]) on synthetic Int64 x = splitmix64(42+i) % 1000 (BASELINE configs[1]); "
                       "rows resident in HBM, row-range shards, no collective", "int64", 8.0, 8.0,
                       "single-pass predicate + ordered compaction"),
    "and2_nulls": ("filter((f > 0.5) AND (x < 200)).select([f, x]) on nullable Float64 f = splitmix64(43+i)>>11 * 2^-53 and nullable "
                   "Int64 x = splitmix64(42+i) % 1000, 5 % nulls each, RV_NULL_DROPS (BASELINE configs[2]); rows resident in HBM",
                   "f64+int64", 16.25, 16.0, "lane-form predicate over two nullable columns + ordered compaction"),
    "filter_agg": ("filter(x > 899) + global SUM(x), COUNT(*) on synthetic Int64 x = splitmix64(42+i) % 1000 (BASELINE configs[4]); "
                   "per-rank partials, one RCCL all-reduce of 2 x int64", "int64", 8.0, 0.0, "read-only masked reduction"),
    # SURVEY 8(f) row 2 across N devices (rv_group_filter_project_host): PCIe-bound, a metric of its own -- never the headline
    "host_table": ("filter(x > 899).select([x]) over a HOST-resident (pinned) Int64 table x = splitmix64(42+i) % 1000: cut into N row "
                   "ranges, every range streamed through its device's own double-buffered chunk pipeline, survivors gathered in rank "
                   "order (StreamingPhysicalPlan::collect(), streaming.rs:71-133 / :343-352)", "int64", 8.0, 8.0,
                   "chunked upload on a second stream overlapped with the fused pass"),
}
HOST_TABLE_ROWS_PER_GPU = 250_000_000  # 2 GB of pinned host memory per GPU


def cpu_baseline():
    """The reference's CPU path for this query is the eager LazyFrame::collect()
    (src/logical_plan/builder.rs:96-104 -> physical_plan/plan.rs:97-150, :68-96); the
    reference is Rust and cannot be built in this pipeline, so the timed code is the C++
    restatement in oracle/ (kind "port"), single-threaded like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    # bounded sample: a 2e6-row probe sizes the run to ~15 s of single-thread work (<= 1e8 rows:
    # the eager path holds several copies of the 32-byte AnyValue cells)
    sample = int(os.environ.get("RV_CPU_SAMPLE_ROWS", 0))
    if sample <= 0:
        probe_sec, _, _ = pyoracle.bench_eager_collect(2_000_000, SEED_X, 1000, LITERAL)
        sample = int(min(100_000_000, max(5_000_000, 15.0 / probe_sec * 2_000_000)))
    sec, rows, _ = pyoracle.bench_eager_collect(sample, SEED_X, 1000, LITERAL)
    s_sec, s_rows, _ = pyoracle.bench_stream(sample, SEED_X, 1000, LITERAL, 1024)
    assert rows == s_rows
    # courtesy figure: typed compress loop on the box's host threads (bounded: 4e8 rows)
    threads = min(os.cpu_count() or 1, 64)
    t_sec, t_rows, _ = pyoracle.bench_threads(400_000_000, SEED_X, 1000, LITERAL, threads)
    return {
        "value": sample / sec,
        "unit": "rows/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample} rows of the same synthetic column; eager collect() restatement over 32-byte AnyValue cells "
                  f"(24-byte String payload + tag: an UPPER bound on the reference's cell -- rustc >= 1.77, which edition 2024 requires, "
                  f"most likely packs the tag into String's capacity niche, 24 bytes), {sec:.2f} s; host has {os.cpu_count()} logical cores",
        "cell_bytes": 32,
        "cell_bytes_note": "upper bound; 24 on toolchains with the capacity niche (SURVEY.md 8a says 24-32, compiler-dependent): the "
                           "timed restatement moves at most a third more bytes per cell than the reference",
        "streaming_restatement_rows_per_s": sample / s_sec,
        "optimised_cpu_rows_per_s": 400_000_000 / t_sec,
        "optimised_cpu_note": f"courtesy: typed compress loop (not the reference's algorithm), {threads} threads, 4e8 rows, "
                              f"{t_sec:.2f} s, survivors {t_rows}",
    }


class Progress:
    """Where a run is and what every rank has reported so far: what an error line is made of.  Every `python bench.py --gpus N`
    ends in ONE JSON line on stdout -- the measurement, or {"error", "phase", per-rank state} with a non-zero exit code (the first
    contact with an 8-GPU node must leave a record, not a traceback: streaming.rs:343-352 is the gather it reports on)."""

    def __init__(self, args):
        self.args, self.phase, self.ranks, self.printed = args, "start", {}, False

    def at(self, phase):
        self.phase = phase
        want = os.environ.get("RV_BENCH_FAIL", "")          # fault injection for the tests: "<rank>:<phase>"
        if want and want.split(":")[1] == phase:
            return int(want.split(":")[0])
        return None

    def error_line(self, message, per_rank=None):
        a = self.args
        return {"error": message, "phase": self.phase, "metric": metric_name(a, a.gpus), "value": None, "unit": "rows/s",
                "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "scaling": a.scaling, "higher_is_better": True,
                "config": {"workload": a.workload, "rows_per_gpu": a.rows if a.scaling == "weak" else None,
                           "global_rows": int(a.global_rows) if a.scaling == "strong" else a.rows * a.gpus},
                "per_rank": per_rank if per_rank is not None else [self.ranks.get(r) for r in range(a.gpus)]}

    def fail(self, message, per_rank=None, code=1):
        if not self.printed:
            print(json.dumps(self.error_line(message, per_rank)), flush=True)
            self.printed = True
        sys.stdout.flush()
        os._exit(code)   # not sys.exit: worker threads / a half-built process group must not hold the exit up


def metric_name(args, world):
    if args.workload != "filter_project" or args.scaling != "weak":
        return f"rows/sec {args.workload}, {args.scaling} scaling (not the headline metric)"
    if world == 1:
        return "rows/sec filter+project, 1e9-row Int64, 10% selectivity"
    return f"rows/sec filter+project, 1e9 Int64 rows PER GPU x {world} GPUs (weak scaling), 10% selectivity"


def survivors_plausible(workload, rows, survivors):
    """The generator is uniform (x = splitmix64 % 1000, f = 53 random bits, 5 % nulls per column), so a shard's survivors are
    binomial around a known p: a rank that filtered the wrong rows, none or twice shows up at once.  Not an exactness check
    (the -m gpu tests compare with the oracle row by row); the bound is 6 sigma."""
    p = {"filter_project": 0.1, "filter_agg": 0.1, "and2_nulls": 0.95 * 0.5 * 0.95 * 0.2}[workload]
    if rows <= 0:
        return survivors == 0
    return abs(survivors - p * rows) <= 6.0 * (rows * p * (1 - p)) ** 0.5 + 2.0


def sampled_window_check(buf, total_rows):
    """The gathered buffer really is the filtered table: windows at the start, the middle (wherever that falls: across
    shard boundaries and, for a 1e10-row table, past 2^32 rows of input) and the end hold only survivors."""
    if total_rows == 0:
        return True
    ok = True
    win = min(4096, total_rows)
    for k in range(8):
        at = (total_rows - win) * k // 7
        w = buf[at:at + win]
        ok = ok and bool((w > LITERAL).all()) and bool((w < 1000).all())
    return ok


def traffic_for(kernel, workload, args):
    """PMC traffic of this kernel from profiles/traffic.json -- attached only when that file's entry was measured on the
    SAME kernel instantiation as the one this run launched, at the same size."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not (os.path.exists(tpath) and args.scaling == "weak" and args.rows == ROWS_PER_GPU and kernel):
        return None, None
    try:
        entry = json.load(open(tpath)).get(workload, {})
        keys = {k.replace("void ", "").replace("rvk::", "").replace(" ", "") for k in (entry.get("fetch_size_raw_avg_per_kernel") or {})}
        if kernel not in keys:
            return None, (f"profiles/traffic.json holds PMC traffic for {sorted(keys)}, this run launched {kernel}: not attached")
        return entry.get("hbm_bytes_per_launch"), (
            f"constant from profiles/traffic.json, measured on this kernel ({kernel}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
            "this same command (tools/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes -- not measured in this run")
    except Exception:  # noqa: BLE001
        return None, None


def make_line(args, world, n_global, elapsed, kernel_ms_avg_max, total_survivors, kernel, end_to_end, hbm_in_use, capi, extra=None, rows_per_launch=None):
    desc, dtype, bytes_per_row, written_per_survivor, kernel_does = WORKLOADS[args.workload]
    ms_per_step = elapsed / args.steps * 1e3
    value = n_global * args.steps / elapsed
    selectivity = total_survivors / n_global if n_global else 0.0
    rows_max = max(capi.shard_range(n_global, world, r)[1] - capi.shard_range(n_global, world, r)[0] for r in range(world))
    rows_launch = rows_per_launch or rows_max            # --seam: a launch is one window of RecordBatches, not the rank's whole table
    algo_read = bytes_per_row * rows_launch              # bytes/row read once x rows of one launch
    algo_total = (bytes_per_row + written_per_survivor * selectivity) * rows_launch  # + compacted survivors written
    achieved = algo_read / (kernel_ms_avg_max * 1e-3) / 1e9 if kernel_ms_avg_max > 0 else 0.0
    traffic, traffic_src = traffic_for(kernel, args.workload, args)
    if rows_launch != rows_max:
        traffic, traffic_src = None, "not attached: profiles/traffic.json was measured on whole-table launches, this run launches windows"
    line = {
        "metric": metric_name(args, world),
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {
            "workload": desc + ("" if args.scaling == "weak" else
                                f"; ONE {n_global:.3g}-row table cut into {world} row range(s) (BASELINE configs[3] / configs[4])"),
            "rows_per_gpu": rows_max,
            "global_rows": n_global,
            "selectivity": selectivity,
            "parallelism": f"row-range x{world}",
            "hbm_in_use_bytes_rank0": hbm_in_use,
            "options": os.environ.get("RV_OPTIONS"),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"{kernel} ({kernel_does})" if kernel else None,
            "kernel_id": kernel,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the same share by the WALL clock of a step (launch overheads, read-backs and -- with --seam -- everything between the windows
            # included): what the rows/s of `value` amount to on the algorithmic bytes
            "frac_by_ms_per_step": bytes_per_row * rows_max / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_per_step > 0 else 0.0,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel_ms_avg": kernel_ms_avg_max,
            "algorithmic_bytes_per_launch": algo_read,
            "achieved_incl_writes": algo_total / (kernel_ms_avg_max * 1e-3) / 1e9 if kernel_ms_avg_max > 0 else 0.0,
        },
        "kernel_only": {"ms_per_step": ms_per_step, "value": value, "unit": "rows/s",
                        "note": "the timed region: outputs stay resident in HBM (== `value`)"},
        "end_to_end": end_to_end,
    }
    if extra:
        line.update(extra)
    return line


def main_host_table(args):
    """--workload host_table: the single-process group driver over a table in pinned HOST memory (any N, also 1)."""
    prog = Progress(args)
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import numpy as np
        from rivulus_amd import capi
        from rivulus_amd.capi import Column, Predicate, Term
        world = args.gpus
        rehearsal = os.environ.get("RV_BENCH_ONE_DEVICE") == "1"
        devices = [0] * world if rehearsal else list(range(world))
        for r in range(world):
            prog.ranks[r] = {"rank": r, "device": devices[r]}
        rows = min(args.rows, HOST_TABLE_ROWS_PER_GPU) * world
        prog.at("group_create")
        group = capi.Group(devices)
        ctxs = [group.context(r) for r in range(world)]
        prog.at("generate")
        xs = ctxs[0].pinned_array(np.int64, rows)   # pinned: chunk uploads run at link rate, truly asynchronous
        step_rows = 1 << 24
        with np.errstate(over="ignore"):
            for o in range(0, rows, step_rows):   # splitmix64(42 + i) % 1000, the generator of SURVEY 8(d), in numpy
                z = np.arange(o, min(rows, o + step_rows), dtype=np.uint64) + np.uint64(SEED_X) + np.uint64(0x9E3779B97F4A7C15)
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                xs[o:o + len(z)] = ((z ^ (z >> np.uint64(31))) % np.uint64(1000)).astype(np.int64)
        x = Column.from_numpy(xs)
        pred = Predicate([Term(0, ">", LITERAL)])
        bad = prog.at("warmup")
        if bad is not None:
            ctxs[bad].set_option("inject_failure", 1)
        survivors, gbs = 0, [0.0] * world
        for _ in range(max(1, args.warmup)):
            res, survivors, gbs = group.filter_project_host([x], pred, [0])
            res.free()
        bad = prog.at("timed")
        if bad is not None:
            ctxs[bad].set_option("inject_failure", 1)
        for c in ctxs:
            c.synchronize()
        steps = max(1, min(args.steps, 5))
        t0 = time.perf_counter()
        all_gbs = []
        for k in range(steps):
            res, survivors, gbs = group.filter_project_host([x], pred, [0])
            all_gbs.append(gbs)
            if k + 1 < steps:
                res.free()
        for c in ctxs:
            c.synchronize()
        elapsed = time.perf_counter() - t0
        got = res.column(0)
        ok = got.length == survivors and sampled_window_check(got.values, got.length) and survivors == int((xs > LITERAL).sum())
        res.free()
        prog.at("report")
        per_rank_gbs = [sum(g[r] for g in all_gbs) / len(all_gbs) for r in range(world)]
        for r in range(world):
            prog.ranks[r].update({"rows": capi.shard_range(rows, world, r)[1] - capi.shard_range(rows, world, r)[0], "upload_gb_s": per_rank_gbs[r]})
        desc, dtype, bytes_per_row, _, kernel_does = WORKLOADS["host_table"]
        ms_per_step = elapsed / steps * 1e3
        line = {
            "metric": f"rows/sec filter+project over a host-resident table, {world} GPU(s) (PCIe-bound; not the headline metric)",
            "value": rows * steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows // world, "global_rows": rows, "selectivity": survivors / rows if rows else 0.0,
                       "parallelism": f"row-range x{world}", "chunk_rows": 1 << 25},
            "pcie": {"per_rank_upload_gb_s": per_rank_gbs, "sum_gb_s": sum(per_rank_gbs),
                     "whole_job_gb_s": bytes_per_row * rows * steps / elapsed / 1e9,
                     "note": "input bytes of a rank's row range over its own filter time (upload of chunk k + 1 overlapped with the pass over chunk k); "
                             "the whole-job figure includes the rank-order gather of the survivors into pinned host memory"},
            "roofline": {"bound": "hbm", "kernel": f"{ctxs[0].last_kernel()} ({kernel_does})", "achieved": bytes_per_row * rows * steps / elapsed / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_per_row * rows * steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "this workload is bound by the host link (see `pcie`), not by HBM: the fraction of the HBM roofline is reported for the contract's sake"},
            "check": "gathered survivors == numpy's count, sampled windows hold only survivors: " + ("ok" if ok else "FAILED"),
            "driver": f"single process, rv_group over devices {devices}" + (" (REHEARSAL on one device: not a scaling measurement)" if rehearsal else ""),
            "per_rank": [prog.ranks[r] for r in range(world)],
        }
        print(json.dumps(line), flush=True)
        prog.printed = True
        group.close()
        if not ok:
            sys.exit(1)
    except SystemExit:
        raise
    except BaseException as ex:  # noqa: BLE001
        prog.fail(f"{type(ex).__name__}: {ex}")


# =====================================================================================================================
# ONE process, N GPUs: rv_group_* (csrc/group.hip).  No torch, no launcher, no process group: the barrier of the
# contract is the join of the N worker threads inside every rv_group_* call plus rv_ctx_synchronize on every context.
# =====================================================================================================================
def main_group(args):
    prog = Progress(args)
    try:
        _main_group(args, prog)
    except SystemExit:
        raise
    except BaseException as ex:  # noqa: BLE001  (every failure leaves a line: RV_ERR_OOM of one rank, a failed pin, an RCCL init error ...)
        prog.fail(f"{type(ex).__name__}: {ex}")


def _main_group(args, prog):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from rivulus_amd import capi
    from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec

    world = args.gpus
    rehearsal = os.environ.get("RV_BENCH_ONE_DEVICE") == "1"   # every rank on device 0 (one-GPU box); never for reported numbers
    devices = [0] * world if rehearsal else list(range(world))
    for r in range(world):
        prog.ranks[r] = {"rank": r, "device": devices[r]}
    n_global = int(args.global_rows) if args.scaling == "strong" else args.rows * world
    prog.at("group_create")
    group = capi.Group(devices)
    ctxs = [group.context(r) for r in range(world)]

    def note_ranks():
        """per-rank state for the line (or the error line): best effort, never raises"""
        for r, c in enumerate(ctxs):
            st = prog.ranks[r]
            try:
                st["rows"] = capi.shard_range(n_global, world, r)[1] - capi.shard_range(n_global, world, r)[0]
                st["survivors"] = c.get_option("last_rows_out")
                st["hbm_bytes"], st["hbm_free_bytes"] = c.device_info().get("hbm_bytes"), c.get_option("hbm_free_bytes")
            except Exception as ex:  # noqa: BLE001
                st["state_error"] = f"{type(ex).__name__}: {ex}"

    def inject(phase):
        bad = prog.at(phase)
        if bad is not None:   # fault injection (tests): the rank's next query call fails inside the library
            ctxs[bad].set_option("inject_failure", 1)

    try:
        inject("generate")
        x = group.generate(synth_spec(RV_INT64, seed=SEED_X, length=n_global,
                                      validity_seed=45 if args.workload == "and2_nulls" else None))
        cols, proj = [x], [0]
        pred = Predicate([Term(0, ">", LITERAL)])
        if args.workload == "and2_nulls":
            f = group.generate(synth_spec(RV_FLOAT64, seed=43, length=n_global, validity_seed=44))
            cols, proj = [f, x], [0, 1]
            pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])

        def sync_all():
            for c in ctxs:
                c.synchronize()

        def step(use_cols=None):
            """One pass of the hot path on every rank at once; the outputs stay in HBM."""
            if args.workload == "filter_agg":
                _, _, cnt = group.filter_agg(cols, pred, 0)   # per-rank partials, then the RCCL all-reduce of 16 bytes
                return cnt
            res, rows = group.filter_project_resident(use_cols or cols, pred, proj)
            res.free()
            return rows

        def timed(use_cols, steps):
            sync_all()
            t0 = time.perf_counter()
            s = 0
            for _ in range(steps):
                s = step(use_cols)
            sync_all()
            return time.perf_counter() - t0, s

        inject("warmup")
        survivors = 0
        for _ in range(args.warmup):
            survivors = step()
        if args.min_seconds > 0:   # --min-seconds: enough steps for a sampler that polls once a second to see the GPU busy
            probe, _ = timed(None, 2)
            args.steps = max(args.steps, int(args.min_seconds / max(probe / 2, 1e-6)) + 1)
        for c in ctxs:
            c.set_option("profile_kernels", 1)
            c.kernel_stats(reset=True)
        inject("timed")
        elapsed, survivors = timed(None, args.steps)
        per_rank = [c.kernel_stats() for c in ctxs]
        for c in ctxs:
            c.set_option("profile_kernels", 0)
        kernel_ms = [ms / max(1, n) for ms, n in per_rank]
        kernel_ms_avg_max = max(kernel_ms)
        kernel = ctxs[0].last_kernel()
        note_ranks()
        for r in range(world):
            prog.ranks[r]["kernel_ms"] = kernel_ms[r]
            if args.workload != "filter_agg":
                prog.ranks[r]["survivors_plausible"] = survivors_plausible(args.workload, prog.ranks[r]["rows"], prog.ranks[r]["survivors"])
        if args.workload != "filter_agg" and not all(prog.ranks[r]["survivors_plausible"] for r in range(world)):
            prog.phase = "survivor check"
            raise RuntimeError("a rank's survivor count is not what the generator's distribution allows (6 sigma): see per_rank")

        # the BASELINE metric as worded ("1e9-row ... at 1/2/4/8 GPUs") read as STRONG scaling: ONE 1e9-row table cut into N row
        # ranges (1.25e8 rows per GPU at N = 8: launch-latency territory) -- a secondary figure next to the weak-scaling `value`
        strong = None
        if args.workload == "filter_project" and args.scaling == "weak" and world > 1 and not args.no_strong_1e9:
            prog.at("strong_1e9")
            n1 = args.rows
            x1 = group.generate(synth_spec(RV_INT64, seed=SEED_X, length=n1))
            for _ in range(2):
                step([x1])
            k1 = max(args.steps, 20)
            e1, s1 = timed([x1], k1)
            x1.free()
            strong = {"value": n1 * k1 / e1, "unit": "rows/s", "ms_per_step": e1 / k1 * 1e3, "steps": k1, "global_rows": n1,
                      "rows_per_gpu": -(-n1 // world), "survivors": s1,
                      "note": f"ONE {n1:.3g}-row table cut into {world} row ranges (strong scaling of the headline size); wall clock "
                              "around K steps, all ranks at once"}

        end_to_end = None
        gathered = survivors * 8 * len(proj)
        want_e2e = not args.no_end_to_end and (args.scaling == "strong" or args.end_to_end or gathered <= (16 << 30))
        if args.workload != "filter_agg" and want_e2e:
            prog.at("end_to_end")
            try:
                e_steps = max(1, min(args.steps, 5))
                filt, gath = [], []

                def e2e_step():
                    res, rows = group.filter_project(cols, pred, proj)   # filter + rank-order gather into pinned host memory
                    return res, rows
                res, _ = e2e_step()   # pins the host buffers once
                res.free()
                sync_all()
                t1 = time.perf_counter()
                for k in range(e_steps):
                    res, rows = e2e_step()
                    st = res.stats()
                    filt.append(st["filter_ms"])
                    gath.append(st["gather_ms"])
                    if k + 1 < e_steps:   # the last result is kept for the check below
                        res.free()
                sync_all()
                e_elapsed = time.perf_counter() - t1
                ok = True
                if args.workload == "filter_project":
                    import ctypes
                    import numpy as np
                    v = capi.RvColumn()
                    capi._check(capi.load().rv_gather_column(res.handle, 0, ctypes.byref(v), None))
                    buf = np.ctypeslib.as_array(ctypes.cast(v.values, ctypes.POINTER(ctypes.c_int64)), (max(1, int(v.length)),))
                    ok = int(v.length) == rows and sampled_window_check(buf, int(v.length))
                res.free()
                end_to_end = {
                    "ms_per_step": e_elapsed / e_steps * 1e3,
                    "value": n_global * e_steps / e_elapsed,
                    "unit": "rows/s",
                    "steps": e_steps,
                    "gathered_bytes_per_step": gathered,
                    "filter_ms": sum(filt) / len(filt),
                    "gather_ms": sum(gath) / len(gath),
                    "note": "rv_group_filter_project: one pass on every rank + device-to-host copy of every rank's survivors into its "
                            "slice of ONE pinned host buffer (rank order == row order; PCIe-bound); sampled-window check of the "
                            "gathered values: " + ("ok" if ok else "FAILED"),
                }
            except Exception as ex:  # noqa: BLE001  (pinning gigabytes of host memory can fail on a small box: the kernel-only line stands)
                end_to_end = {"error": f"{type(ex).__name__}: {ex}"}

        prog.at("report")
        hbm = ctxs[0].device_info().get("hbm_bytes")
        extra = {
            "driver": f"single process, rv_group over devices {devices}: one context + one host thread per device"
                      + (" (REHEARSAL on one device: not a scaling measurement)" if rehearsal else ""),
            "rccl_ranks": group.stat("rccl_ranks"),
            "per_rank": [prog.ranks[r] for r in range(world)],
            "cpu_baseline": {"see": "the N = 1 line (`python bench.py --gpus 1`): the CPU baseline is timed there only, on rank 0, as the "
                                    "contract asks; it does not depend on N"},
        }
        if strong is not None:
            extra["strong_1e9"] = strong
        if args.workload == "filter_agg":
            extra["allreduce"] = {"calls": group.stat("allreduce_calls"), "last_filter_phase_us": group.stat("last_agg_filter_us"),
                                  "last_allreduce_phase_us": group.stat("last_allreduce_us"),
                                  "path": "RCCL ncclAllReduce over the group's devices" if group.stat("rccl_ranks") else
                                          "host sum in rank order (a device listed twice cannot form an RCCL communicator)"}
        line = make_line(args, world, n_global, elapsed, kernel_ms_avg_max, float(survivors), kernel, end_to_end, None, capi, extra)
        line["config"]["hbm_bytes_device0"] = hbm
        print(json.dumps(line), flush=True)
        prog.printed = True
    except BaseException:
        note_ranks()
        raise
    for c in cols:
        c.free()
    group.close()


class HostGather:
    """The rank-order gather of configs[3] (streaming.rs:343-352 with the shards as the batches): ONE pinned host
    buffer; rank r copies its survivors to rows [prefix_r, prefix_r + rows_r).  With one process per GPU the buffer
    is a POSIX shared-memory segment every rank maps and pins (rv_host_register)."""

    def __init__(self, ctx, capi, dist, rank, world, total_rows, ncols):
        import numpy as np
        from multiprocessing import shared_memory
        self.ctx, self.capi, self.dist, self.rank, self.world = ctx, capi, dist, rank, world
        self.np = np
        self.total_rows, self.ncols = total_rows, ncols
        nbytes = max(8, total_rows * 8 * ncols)
        self.shm = None
        if world == 1:
            self.buf = ctx.pinned_array(np.int64, max(1, total_rows * ncols))
        else:
            # /dev/shm is a tmpfs: a segment larger than what it can back would fault (SIGBUS) on first touch, which no
            # try/except catches -- check the room first and let every rank agree on the outcome
            name = [None]
            if rank == 0:
                st = os.statvfs("/dev/shm")
                if st.f_bavail * st.f_frsize > nbytes + (nbytes >> 2) + (64 << 20):
                    self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
                    name[0] = self.shm.name
                else:
                    name[0] = f"!/dev/shm has {st.f_bavail * st.f_frsize} bytes free, the gather needs {nbytes}"
            dist.broadcast_object_list(name, src=0)
            if name[0].startswith("!"):
                raise RuntimeError(name[0][1:])
            if rank != 0:
                self.shm = shared_memory.SharedMemory(name=name[0])
            self.buf = np.frombuffer(self.shm.buf, dtype=np.int64, count=max(1, total_rows * ncols))
            ok = [False] * world
            try:
                capi._check(capi.load().rv_host_register(ctx.handle, self.buf.ctypes.data, nbytes))
                mine = True
            except Exception:  # noqa: BLE001
                mine = False
            dist.all_gather_object(ok, mine)
            if not all(ok):  # every rank leaves together, the segment does not outlive the attempt
                if mine:
                    capi.load().rv_host_unregister(ctx.handle, self.buf.ctypes.data)
                del self.buf
                try:
                    self.shm.close()
                except BufferError:
                    pass
                if rank == 0:
                    self.shm.unlink()
                self.shm = None
                raise RuntimeError("pinning the shared gather buffer failed on some rank")

    def put(self, outs, prefix_rows):
        """D2H of this rank's output columns into its slice of the shared buffer (column-major: column j at j * total)."""
        import ctypes
        lib = self.capi.load()
        has = ctypes.c_int()
        for j, o in enumerate(outs):
            dst = self.buf.ctypes.data + 8 * (j * self.total_rows + prefix_rows)
            self.capi._check(lib.rv_download(self.ctx.handle, o.handle, ctypes.c_void_p(dst), None, ctypes.byref(has)))

    def close(self):
        if self.shm is not None:
            self.capi.load().rv_host_unregister(self.ctx.handle, self.buf.ctypes.data)
            del self.buf
            try:
                self.shm.close()
            except BufferError:  # a view of the mapping is still alive somewhere: the mapping goes with the process
                pass
            if self.rank == 0:
                self.shm.unlink()


# =====================================================================================================================
# one process per GPU (N = 1, or N > 1 under torch.distributed.run)
# =====================================================================================================================
def main_ranks(args):
    """One process per GPU.  With N > 1 every phase runs under a guard that CATCHES a rank's failure instead of raising it, and the
    ranks agree on the outcome at the end of every phase (one all_gather of {ok, error, state}): a rank that failed keeps walking
    to the next agreement point -- through the barriers the others stand in -- so nobody is left waiting in a collective, rank 0
    prints the error line with every rank's state, and every rank exits non-zero.  (A rank that dies outright -- a kill, a
    segfault -- takes the launcher down with it: rank 0's SIGTERM handler then leaves the line.)"""
    # read by the HSA runtime when it initialises (first GPU call): must be in the environment before torch touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import signal
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    prog = Progress(args)
    me = {"rank": rank, "device": local_rank}
    prog.ranks[rank] = me
    if rank == 0 and world > 1:   # the launcher ends the surviving ranks with SIGTERM when one of them died
        signal.signal(signal.SIGTERM, lambda *_: prog.fail("terminated by the launcher: another rank died", code=143))
    # RV_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 protocol on a box with a single GPU (every rank on
    # device 0, gloo instead of RCCL); never used for reported numbers.
    rehearsal = os.environ.get("RV_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
        me["device"] = 0
    dist = None
    failure = []   # this rank's first failure: [phase, message]

    def guard(phase, fn, *a):
        """run one phase's work; a failure is recorded, not raised (agree() below settles it with the other ranks)"""
        bad = prog.at(phase)
        if failure:
            return None
        try:
            if bad == rank:
                raise RuntimeError(f"injected failure on rank {rank} in phase {phase}")
            return fn(*a)
        except BaseException as ex:  # noqa: BLE001
            failure[:] = [phase, f"{type(ex).__name__}: {ex}"]
            return None

    def agree():
        """every rank calls this at the end of every phase: all fine, or one error line and a non-zero exit everywhere"""
        if dist is None:
            if failure:
                prog.phase = failure[0]
                prog.fail(failure[1], [me])
            return
        got = [None] * world
        dist.all_gather_object(got, {"failure": list(failure), "state": me})
        bad = [g for g in got if g["failure"]]
        if bad:
            prog.phase = bad[0]["failure"][0]
            msg = "; ".join(f"rank {g['state']['rank']}: {g['failure'][1]}" for g in bad)
            if rank == 0:
                prog.fail(msg, [g["state"] for g in got])
            os._exit(1)

    try:
        torch.cuda.set_device(local_rank)
        if world > 1:
            import torch.distributed as dist
            prog.at("init_process_group")
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    except BaseException as ex:  # noqa: BLE001  (no process group: nothing to agree through -- every rank reports for itself)
        prog.fail(f"rank {rank}: {type(ex).__name__}: {ex}", [me])
    red_dev = "cpu" if (rehearsal or dist is None) else "cuda"

    from rivulus_amd import capi
    from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec

    n_global = int(args.global_rows) if args.scaling == "strong" else args.rows * world
    begin, end = capi.shard_range(n_global, world, rank)
    rows_here = end - begin
    me["rows"] = rows_here
    S = {"ctx": None, "x": None, "f": None, "comm": None, "prepared": None, "seam": None}
    seam_form, seam_rows = None, 0
    if args.seam:
        seam_form, _, r = args.seam.partition(":")
        seam_rows = int(r or 1024)
        if seam_form not in ("chunked", "handles") or args.workload == "filter_agg":
            prog.fail("--seam takes chunked:<rows per batch> or handles:<rows per batch>, for the filter + project workloads", [me])
    SEAM_WINDOW = 1 << 28  # rows a stream operator looks ahead by (tools/batch_sweep.py: 262 144 batches of 1024 rows)
    pred = Predicate([Term(0, ">", LITERAL)])
    pred3 = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])

    def note_state():
        try:
            if S["ctx"] is not None:
                me["hbm_free_bytes"] = S["ctx"].get_option("hbm_free_bytes")
                me["hbm_bytes"] = S["ctx"].device_info().get("hbm_bytes")
        except Exception:  # noqa: BLE001
            pass

    def setup():
        S["ctx"] = capi.Context(local_rank)
        S["x"] = S["ctx"].generate(synth_spec(RV_INT64, seed=SEED_X, length=rows_here, first_row=begin,
                                              validity_seed=45 if args.workload == "and2_nulls" else None))
        if args.workload == "and2_nulls":
            S["f"] = S["ctx"].generate(synth_spec(RV_FLOAT64, seed=43, length=rows_here, first_row=begin, validity_seed=44))
        # the query's argument structs are built once, as a caller that repeats a query builds them (capi.prepared_filter_project)
        if args.workload == "and2_nulls":
            S["prepared"] = S["ctx"].prepared_filter_project([S["f"], S["x"]], pred3, [0, 1])
        elif args.workload == "filter_project":
            S["prepared"] = S["ctx"].prepared_filter_project([S["x"]], pred, [0])
        if seam_form:
            # the workload "fed through the stream seam" -- FilterStream over MemoryStream at the reference's batch size
            # (stream.rs:58-163), windows of 2^28 rows, two in flight (rv_filter_project_chunked_begin / _batches_begin + _window_finish)
            import numpy as np
            cols = [S["f"], S["x"]] if args.workload == "and2_nulls" else [S["x"]]
            nwin = (rows_here + SEAM_WINDOW - 1) // SEAM_WINDOW
            tables = [[c.slice(w * SEAM_WINDOW, min(SEAM_WINDOW, rows_here - w * SEAM_WINDOW)) for c in cols] for w in range(nwin)]
            batch_lists = [[[c.slice(o, min(seam_rows, t[0].length - o)) for c in t] for o in range(0, t[0].length, seam_rows)] for t in tables] if seam_form == "handles" else []
            S["seam"] = {"tables": tables, "batches": batch_lists, "handles": [S["ctx"].batch_handles(bl) for bl in batch_lists],
                         "counts": [S["ctx"].pinned_array(np.uint64, (SEAM_WINDOW + seam_rows - 1) // seam_rows) for _ in range(2)],
                         "pred": pred3 if args.workload == "and2_nulls" else pred, "proj": [0, 1] if args.workload == "and2_nulls" else [0]}
    guard("generate", setup)
    note_state()
    agree()
    ctx, x, f, prepared = S["ctx"], S["x"], S["f"], S["prepared"]

    if args.workload == "filter_agg" and world > 1 and not rehearsal:
        uid = [None]
        guard("comm_init", lambda: uid.__setitem__(0, capi.comm_unique_id() if rank == 0 else None))
        agree()
        dist.broadcast_object_list(uid, src=0)
        guard("comm_init", lambda: S.__setitem__("comm", capi.Comm(ctx, uid[0], world, rank)))
        agree()
    comm = S["comm"]

    def query():
        """One pass of the hot path over this rank's rows; the outputs stay in HBM."""
        if args.workload == "filter_agg":
            s, _, c = ctx.filter_agg([x], pred, 0)
            me["survivors"] = c
            if comm is not None:
                s, c = comm.allreduce_sum_count(s, c)   # every rank ends with the global 16 bytes
            elif dist is not None:                      # rehearsal on one device: the same payload over gloo
                t2 = torch.tensor([s, c], dtype=torch.int64)
                dist.all_reduce(t2)
                s, c = int(t2[0]), int(t2[1])
            return [], c, s
        outs, rows = prepared(keep=True)
        return outs, rows, None

    def seam_steps(steps):
        """`steps` passes over the rank's rows through seam S1, as ONE stream of windows (a stream operator does not stop between two
        passes of a benchmark): window i + 1 is begun before window i is finished; the per-batch survivor counts land in the operator's
        pinned arrays; every window's outputs are released at once.  Returns the survivors of the last pass."""
        q = S["seam"]
        nwin = len(q["tables"])

        def begin(i):
            w = i % nwin
            if seam_form == "handles":
                return ctx.window_begin(q["pred"], q["proj"], q["counts"][i % 2], handles=q["handles"][w])
            return ctx.window_begin(q["pred"], q["proj"], q["counts"][i % 2], cols=q["tables"][w], chunk_rows=seam_rows)
        total, pending, last = 0, [begin(0)], 0
        for i in range(steps * nwin):
            if i + 1 < steps * nwin:
                pending.append(begin(i + 1))
            outs, _, _, tot = pending.pop(0)(False)
            total += tot
            for o in outs:
                o.free()
            if (i + 1) % nwin == 0:
                last, total = total, 0
        return last

    def seam_step():
        return seam_steps(1)

    def step():
        """One step of the timed region: one pass (descriptor memset + fused kernel + 512-byte read-back); the outputs are
        released at once."""
        if S["seam"] is not None:
            return seam_step()
        if prepared is not None:
            return prepared()[1]
        return query()[1]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    R = {"survivors": 0, "elapsed": 0.0}

    def warm():
        for _ in range(args.warmup):
            R["survivors"] = step()
    # (filter_agg with a collective inside every step: a rank that fails mid-loop leaves the others in the all-reduce until the
    # communicator's own timeout -- the injected failures of the tests fire before a loop starts)
    guard("warmup", warm)
    agree()
    if args.min_seconds > 0:   # --min-seconds: enough steps for a sampler that polls once a second to see the GPU busy
        probe = [1e-3]

        def probe_run():
            torch.cuda.synchronize()
            t = time.perf_counter()
            step(), step()
            torch.cuda.synchronize()
            probe[0] = (time.perf_counter() - t) / 2
        guard("warmup", probe_run)
        agree()
        t = torch.tensor([probe[0]], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        args.steps = max(args.steps, int(args.min_seconds / max(float(t.item()), 1e-6)) + 1)

    def timed_loop(steps):
        if S["seam"] is not None:   # K passes as one stream of windows (two in flight throughout)
            R["survivors"] = seam_steps(steps)
            return
        for _ in range(steps):
            R["survivors"] = step()

    guard("timed", lambda: (ctx.set_option("profile_kernels", 1), ctx.kernel_stats(reset=True)))
    barrier()
    t0 = time.perf_counter()
    guard("timed", timed_loop, args.steps)
    barrier()
    R["elapsed"] = time.perf_counter() - t0
    survivors, elapsed = R["survivors"], R["elapsed"]
    K = {"ms": 0.0, "n": 1, "kernel": None}

    def stats():
        K["ms"], K["n"] = ctx.kernel_stats()
        ctx.set_option("profile_kernels", 0)
        K["kernel"] = ctx.last_kernel()
    guard("timed", stats)
    kernel_ms, launches, kernel = K["ms"], K["n"], K["kernel"]
    me["kernel_ms"] = kernel_ms / max(1, launches)
    if args.workload != "filter_agg":
        me["survivors"] = survivors
    if not failure and args.workload != "filter_agg" or (args.workload == "filter_agg" and "survivors" in me):
        me["survivors_plausible"] = survivors_plausible(args.workload, rows_here, me.get("survivors", 0))
        if not me["survivors_plausible"] and not failure:
            failure[:] = ["survivor check", f"{me.get('survivors')} survivors of {rows_here} rows is not what the generator's distribution allows (6 sigma)"]
    note_state()
    agree()

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    prog.at("reduce")
    elapsed = reduce_max(elapsed)
    kernel_ms_avg_max = reduce_max(kernel_ms / max(1, launches))
    if args.workload == "filter_agg" and world > 1:
        total_survivors = float(survivors)  # already global (all-reduced)
    else:
        tot = torch.tensor([float(survivors)], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_survivors = float(tot.item())
    per_rank = [me]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)

    # ---- the BASELINE metric as worded ("1e9-row ... at 1/2/4/8 GPUs") read as STRONG scaling: ONE 1e9-row table cut into N
    #      row ranges (1.25e8 rows per GPU at N = 8: launch-latency territory), next to the weak-scaling `value` ------------------
    strong = None
    if args.workload == "filter_project" and args.scaling == "weak" and world > 1 and not args.no_strong_1e9:
        n1 = args.rows
        b1, e1 = capi.shard_range(n1, world, rank)
        k1 = max(args.steps, 20)
        T = {"elapsed": 0.0, "rows": 0}

        def strong_setup():
            S["x1"] = ctx.generate(synth_spec(RV_INT64, seed=SEED_X, length=e1 - b1, first_row=b1))
            S["p1"] = ctx.prepared_filter_project([S["x1"]], pred, [0])
            S["p1"](), S["p1"]()
        guard("strong_1e9", strong_setup)
        barrier()
        t1 = time.perf_counter()

        def strong_loop():
            for _ in range(k1):
                T["rows"] = S["p1"]()[1]
        guard("strong_1e9", strong_loop)
        barrier()
        T["elapsed"] = time.perf_counter() - t1
        guard("strong_1e9", lambda: S["x1"].free())
        agree()
        e_max = reduce_max(T["elapsed"])
        tot1 = torch.tensor([float(T["rows"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot1, op=dist.ReduceOp.SUM)
        strong = {"value": n1 * k1 / e_max, "unit": "rows/s", "ms_per_step": e_max / k1 * 1e3, "steps": k1, "global_rows": n1,
                  "rows_per_gpu": -(-n1 // world), "survivors": int(tot1.item()),
                  "note": f"ONE {n1:.3g}-row table cut into {world} row ranges (strong scaling of the headline size); barrier + "
                          "synchronize on both sides, max over ranks"}

    # ---- end to end: + the survivors gathered on the host in rank order, in pinned memory -------
    end_to_end = None
    want_e2e = not args.no_end_to_end and (world == 1 or args.scaling == "strong" or args.end_to_end)
    if args.workload != "filter_agg" and want_e2e:
        prog.at("end_to_end")
        try:
            counts = torch.tensor([survivors], dtype=torch.int64, device=red_dev)
            if dist is not None:
                allc = [torch.zeros_like(counts) for _ in range(world)]
                dist.all_gather(allc, counts)
                allc = [int(c.item()) for c in allc]
            else:
                allc = [survivors]
            ncols_out = 2 if args.workload == "and2_nulls" else 1
            gather = HostGather(ctx, capi, dist, rank, world, sum(allc), ncols_out)
            e_steps = max(1, min(args.steps, 5))

            def e2e_step():
                outs, rows, _ = query()
                if dist is not None:  # the N survivor counts: 8 bytes per rank, then a prefix sum
                    mine = torch.tensor([rows], dtype=torch.int64, device=red_dev)
                    got = [torch.zeros_like(mine) for _ in range(world)]
                    dist.all_gather(got, mine)
                    prefix = sum(int(c.item()) for c in got[:rank])
                else:
                    prefix = 0
                gather.put(outs, prefix)
                for o in outs:
                    o.free()
            e2e_step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(e_steps):
                e2e_step()
            barrier()
            e_elapsed = reduce_max(time.perf_counter() - t1)
            # the gathered buffer really is the filtered table: sampled windows from the start to the end hold survivors only
            ok = sampled_window_check(gather.buf[:sum(allc)], sum(allc)) if args.workload == "filter_project" else True
            end_to_end = {
                "ms_per_step": e_elapsed / e_steps * 1e3,
                "value": n_global * e_steps / e_elapsed,
                "unit": "rows/s",
                "steps": e_steps,
                "gathered_bytes_per_step": sum(allc) * 8 * ncols_out,
                "note": "one pass + device-to-host copy of every rank's survivors into its slice of ONE pinned host buffer "
                        "(rank order == row order; PCIe-bound); sampled-window check of the gathered values: " + ("ok" if ok else "FAILED"),
            }
            gather.close()
        except Exception as ex:  # noqa: BLE001  (pinning gigabytes of host memory can fail on a small box: the kernel-only line stands)
            end_to_end = {"error": f"{type(ex).__name__}: {ex}"}

    prog.at("report")
    try:
        free_b, total_b = torch.cuda.mem_get_info()
        hbm_in_use = int(total_b - free_b)   # inputs + pooled output / scratch blocks of this rank
    except Exception:  # noqa: BLE001
        hbm_in_use = None
    if rank == 0:
        extra = {"driver": "one process per GPU" + ("" if world == 1 else " under torch.distributed.run ("
                                                    + ("gloo, REHEARSAL on one device" if rehearsal else "RCCL") + ")"),
                 "per_rank": per_rank}
        if strong is not None:
            extra["strong_1e9"] = strong
        if seam_form:
            extra["seam"] = {"form": "rv_filter_project_chunked_begin" if seam_form == "chunked" else "rv_filter_project_batches_begin (one handle per batch and column)",
                             "rows_per_batch": seam_rows, "rows_per_window": SEAM_WINDOW, "windows_in_flight": 2,
                             "windows_per_step": (rows_here + SEAM_WINDOW - 1) // SEAM_WINDOW,
                             "steps_run_as": "one stream of steps x windows_per_step windows: the operator does not drain between two passes",
                             "note": "SURVEY 8(d): the workload fed through stream seam S1 at the reference's batch size; `roofline` is per WINDOW launch "
                                     "(kernel_ms_avg = the average window's pass, over rows_per_gpu / windows_per_step rows), `roofline.frac_by_ms_per_step` the whole step by the wall clock"}
        line = make_line(args, world, n_global, elapsed, kernel_ms_avg_max, total_survivors, kernel, end_to_end, hbm_in_use, capi, extra,
                         # (kernel_ms_avg averages over the step's windows, the last one shorter: so does the rows figure)
                         rows_per_launch=rows_here / ((rows_here + SEAM_WINDOW - 1) // SEAM_WINDOW) if seam_form else None)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        elif world > 1:
            line["cpu_baseline"] = {"see": "the N = 1 line (`python bench.py --gpus 1`): the CPU baseline is timed there only, on rank 0, "
                                           "as the contract asks; it does not depend on N"}
        print(json.dumps(line), flush=True)
        prog.printed = True

    if comm is not None:
        comm.close()
    x.free()
    if f is not None:
        f.free()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU under weak scaling (default: the BASELINE size)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --rows per GPU; strong: ONE table of --global-rows rows cut into N row ranges (configs[3]/[4])")
    ap.add_argument("--global-rows", type=float, default=GLOBAL_ROWS_STRONG, help="table size under --scaling strong (default 1e10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the D2H-inclusive figure")
    ap.add_argument("--end-to-end", action="store_true",
                    help="take the D2H-inclusive figure also in a weak-scaling run on several ranks (default there: only with "
                         "--scaling strong -- the leg has collectives of its own, and a rank that fails inside it would leave the "
                         "others waiting)")
    ap.add_argument("--min-seconds", type=float, default=0.0,
                    help="raise --steps so that the timed region lasts at least this long (default 0: exactly --steps); for a "
                         "GPU-busy sampler that polls once a second -- 20 steps of the headline are 27 ms")
    ap.add_argument("--no-strong-1e9", action="store_true",
                    help="skip the secondary strong-scaling figure (ONE 1e9-row table over the N GPUs) of a weak-scaling N > 1 run")
    ap.add_argument("--seam", default=None, metavar="FORM:ROWS",
                    help="feed the workload through the stream seam S1 (SURVEY 8d): chunked:1024 = rv_filter_project_chunked_begin over windows of "
                         "2^28 rows cut into 1024-row RecordBatches, handles:1024 = rv_filter_project_batches_begin over one handle per batch and "
                         "column; two windows in flight, per-batch survivor counts delivered.  One process per GPU only")
    ap.add_argument("--workload", default="filter_project", choices=sorted(WORKLOADS),
                    help="filter_project = BASELINE configs[1] (default, the headline); and2_nulls = configs[2]; "
                         "filter_agg = configs[4] (SUM/COUNT + RCCL all-reduce of 16 bytes)")
    args = ap.parse_args()
    # The launch mode is decided from the environment alone, before anything touches a GPU (and nothing is ever
    # re-executed): a launcher that set RANK / WORLD_SIZE gets one rank per process; `--gpus N` without one gets the
    # single-process driver.
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.workload == "host_table":
        if launched and int(os.environ.get("RANK", "0")) != 0:
            return   # one process drives every device of the node: the launcher's other ranks have nothing to do
        main_host_table(args)
    elif args.gpus > 1 and not launched:
        main_group(args)
    else:
        main_ranks(args)


if __name__ == "__main__":
    main()
