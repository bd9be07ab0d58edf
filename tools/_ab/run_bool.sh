cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shapes_scale_gpu.py tests/test_paths_gpu.py -q -m gpu -x > gpurun_out/bool_tests.log 2>&1; tail -3 gpurun_out/bool_tests.log
python3 tools/batch_sweep.py bool --json gpurun_out/r05_batch_bool.json > gpurun_out/r05_batch_bool.txt 2>&1; tail -6 gpurun_out/r05_batch_bool.txt
