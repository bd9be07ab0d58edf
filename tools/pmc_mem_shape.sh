#!/bin/bash
# HBM traffic counters of the kernels of a tools/shape_run.py shape (own passes, no trace domains; units per MI355X_MICROARCH.md:
# FETCH_SIZE x 1024 x 2 bytes... see tools/summarize_profiles.py).   tools/pmc_mem_shape.sh <shape> [kernel-substring]
set -u
w=${1:-strings_dense}
match=${2:-str}
export TMPDIR=/tmp
out=gpurun_out/pmc_mem_$w
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 tools/shape_run.py $w 2 > $out/$c.log 2>&1 \
    || { echo "pmc_mem_shape.sh: the $c pass failed (rc $?): see $out/$c.log" >&2; tail -5 $out/$c.log >&2; exit 1; }
done
python3 - "$out" "$match" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if sys.argv[2] in k:
            acc[(k.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    gb = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
    print(f"{k:62s} {c:12s} {gb:8.3f} GB per launch")
PY
