cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shapes_scale_gpu.py tests/test_parity_gpu.py tests/test_skew_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x > gpurun_out/wide_tests.log 2>&1; tail -3 gpurun_out/wide_tests.log
for s in wide9n wide9n_half wide9n_dense wide9 bool_c; do python3 tools/shape_run.py $s 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['shape'], round(d['call_ms'],3), 'ms', round(d['frac_of_8TBps_call'],3), d['kernel'])"; done
