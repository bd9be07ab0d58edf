#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run via gpurun from the repo root).
#   tools/profile_round.sh <tag> [workloads...]   -> gpurun_out/prof_<tag>/<workload>/{bench.json,stats/,fetch/,write/} + summary/
# workloads: filter_project (BASELINE configs[1], the headline), and2_nulls (configs[2]), filter_agg (configs[4] per GPU),
#            shape:bool_xb, shape:bool_x, shape:strings, shape:or2 (tools/shape_run.py)
# Counters run in their own passes, never combined with a trace domain; the program itself follows `--`.
set -u
tag=${1:-r02}
shift || true
workloads=${*:-filter_project and2_nulls filter_agg shape:bool_xb shape:strings}
root=$(pwd)
export TMPDIR=/tmp
for w in $workloads; do
  out=$root/gpurun_out/prof_$tag/${w#shape:}
  mkdir -p "$out"
  if [[ $w == shape:* ]]; then
    cmd=(python3 tools/shape_run.py "${w#shape:}" 5)
    short=(python3 tools/shape_run.py "${w#shape:}" 2)
    python3 tools/shape_run.py "${w#shape:}" 5 > "$out/bench.json" 2> "$out/bench.err"
  else
    cmd=(python3 bench.py --workload "$w" --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end)
    short=(python3 bench.py --workload "$w" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end)
    if [[ $w == filter_project ]]; then python3 bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
    else python3 bench.py --workload "$w" --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; fi
  fi
  tail -c 400 "$out/bench.json"; echo
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- "${cmd[@]}" > "$out/stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- "${short[@]}" > "$out/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- "${short[@]}" > "$out/write.log" 2>&1
done
python3 tools/summarize_profiles.py "$root/gpurun_out/prof_$tag" "$tag"
