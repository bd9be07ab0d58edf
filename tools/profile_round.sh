#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run via gpurun from the repo root).
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{bench.json,stats/,fetch/,write/}
# Counters run in their own passes, never combined with a trace domain.
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
tail -c 600 "$out/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/write.log" 2>&1
find "$out" -name "*.csv" | head -20
python3 tools/summarize_profiles.py "$out" "$tag"
