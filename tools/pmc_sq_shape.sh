#!/bin/bash
# SQ counters of the kernels of a tools/shape_run.py shape (diagnostic; own passes, no trace domains).
#   tools/pmc_sq_shape.sh <shape> [kernel-name-substring] -> gpurun_out/pmc_sq_<shape>/
set -u
w=${1:-bool_c}
match=${2:-bits_compact}
export TMPDIR=/tmp
out=gpurun_out/pmc_sq_$w
mkdir -p $out
# a pass that fails must not leave an empty summary behind: each rocprofv3 exit status is checked
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $out/a -- python3 tools/shape_run.py $w 2 > $out/a.log 2>&1 \
  || { echo "pmc_sq_shape.sh: the first counter pass failed (rc $?): see $out/a.log" >&2; tail -5 $out/a.log >&2; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $out/b -- python3 tools/shape_run.py $w 2 > $out/b.log 2>&1 \
  || { echo "pmc_sq_shape.sh: the second counter pass failed (rc $?): see $out/b.log" >&2; tail -5 $out/b.log >&2; exit 1; }
python3 - "$out" "$match" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if sys.argv[2] in k:
            acc[(k.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:62s} {c:24s} {sum(v)/len(v):16.0f}")
PY
