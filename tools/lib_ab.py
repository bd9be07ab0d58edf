"""The headline pass (x > 899 -> [x], 1e9 rows) and config 3 under TWO BUILDS of the library on one box, alternating processes (boxes
differ by +- 3 %, so a before / after needs the same box): kernel time by HIP events and wall per call.
    python3 tools/lib_ab.py tools/_ab/librivulus_gpu_r04.so [rounds]     (the other build: the tree's own)
A build of another commit: `git worktree add X <commit>; make -C X/rivulus_amd/csrc; cp X/rivulus_amd/csrc/librivulus_gpu.so tools/_ab/`."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from rivulus_amd import capi
    from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec
    ctx = capi.Context(0)
    n = 1_000_000_000
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    out = {}
    shapes = [("config2", [x], Predicate([Term(0, ">", 899)]), [0])]
    if len(sys.argv) > 2 and sys.argv[2] == "all":
        f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
        xv = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
        shapes.append(("config3", [f, xv], Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1]))
    for name, cols, pred, proj in shapes:
        run = ctx.prepared_filter_project(cols, pred, proj)
        for _ in range(5):
            run()
        best = (1e9, 1e9)
        for _ in range(3):
            ctx.set_option("profile_kernels", 1)
            ctx.kernel_stats(reset=True)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 20 * 1e3
            ms, k = ctx.kernel_stats()
            ctx.set_option("profile_kernels", 0)
            best = min(best, (ms / 20, wall))
        out[name] = {"kernel_ms": round(best[0], 4), "call_ms": round(best[1], 4), "kernel": ctx.last_kernel()}
    print(json.dumps(out))
    sys.exit(0)

other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "config2"
for r in range(rounds):
    for label, lib in (("other", other), ("tree ", None)):
        env = dict(os.environ)
        if lib:
            env["RIVULUS_GPU_LIB"] = lib
        else:
            env.pop("RIVULUS_GPU_LIB", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", what], env=env, capture_output=True, text=True)
        line = [q for q in p.stdout.splitlines() if q.startswith("{")]
        print(f"round {r} {label} {line[-1] if line else p.stderr[-300:]}", flush=True)
