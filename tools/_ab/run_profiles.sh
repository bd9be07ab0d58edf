cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 filter_project and2_nulls filter_agg shape:iid10 shape:sorted10 shape:clustered10 shape:clustered10k shape:sorted84 shape:iid84 shape:sorted50 shape:iid50 > gpurun_out/profile_round_r05.log 2>&1
tail -40 gpurun_out/profile_round_r05.log
python3 tools/skew_sweep.py > gpurun_out/r05_skew_sweep.txt 2>&1
# the seam: a kernel trace of window calls, config 3 chunked (bench --seam) and the timeline of the last windows
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/seam_and2_chunked/stats -- python3 bench.py --workload and2_nulls --seam chunked:1024 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/prof_r05/seam_and2_chunked_bench.json 2>/dev/null
python3 tools/seam_pipeline.py --timeline gpurun_out/prof_r05/seam_and2_chunked/stats > gpurun_out/r05_seam_timeline.txt 2>&1
python3 bench.py --workload and2_nulls --seam chunked:1024 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_seam_and2_chunked_bench.json 2>/dev/null
python3 bench.py --workload and2_nulls --seam handles:1024 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_seam_and2_handles_bench.json 2>/dev/null
python3 bench.py --seam chunked:1024 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_seam_config2_chunked_bench.json 2>/dev/null
python3 bench.py --seam handles:1024 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_seam_config2_handles_bench.json 2>/dev/null
python3 tools/batch_sweep.py seam bool --json gpurun_out/r05_batch_sweep.json > gpurun_out/r05_batch_sweep.txt 2>&1
tail -8 gpurun_out/r05_batch_sweep.txt
