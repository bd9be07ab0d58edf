cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_group_gpu.py -q -m gpu -x -k "host_table or every_array" > gpurun_out/group_tests.log 2>&1; tail -3 gpurun_out/group_tests.log
python3 bench.py --workload host_table --gpus 1 > gpurun_out/bench_host1.json 2> gpurun_out/bench_host.err; tail -c 900 gpurun_out/bench_host1.json
RV_BENCH_ONE_DEVICE=1 python3 bench.py --workload host_table --gpus 2 --rows 100000000 > gpurun_out/bench_host2.json 2>> gpurun_out/bench_host.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_host2.json') if l.startswith('{')][-1]); print(d['value'], d['pcie'], d['check'])"
tail -3 gpurun_out/bench_host.err
