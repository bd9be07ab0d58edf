cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shapes_scale_gpu.py tests/test_parity_gpu.py -q -m gpu -x -k "seam or batches or chunked or batch or begin or windows" > gpurun_out/seam_tests.log 2>&1; tail -3 gpurun_out/seam_tests.log
for w in config2 config3; do for f in chunked handles; do python3 tools/seam_pipeline.py $w $f 12; done; done
python3 tools/batch_sweep.py seam bool --json gpurun_out/r05_batch_sweep.json > gpurun_out/r05_batch_sweep.txt 2>&1; tail -9 gpurun_out/r05_batch_sweep.txt
