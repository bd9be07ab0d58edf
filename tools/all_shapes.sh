#!/bin/bash
# Every shape of tools/shape_run.py on ONE box, one line each (call = the whole rv_filter_project call): the same-box table under
# profiles/rNN_shapes.txt.      bash tools/all_shapes.sh > gpurun_out/rNN_shapes.txt
echo "# tools/shape_run.py <shape> 8, every shape on ONE box (call = whole rv_filter_project call, HIP-inclusive)"
for s in bool_xb bool_x bool_c bool_c_half bool_c_dense strings strings_dense or2 dense1 dense3 wide5 wide9 wide5_dense wide9_dense wide9n wide9n_half wide9n_dense \
         iid10 iid50 iid84 sorted10 sorted50 sorted84 sorteddesc10 clustered10 clustered10k clustered50 clustered50k; do
  python3 tools/shape_run.py $s 8 2>/dev/null | python3 -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print(f\"{d['shape']:14s} {d['kernel'][:58]:58s} rows {d['rows']:>11d} survivors {d['survivors']:>11d}  call {d['call_ms']:7.3f} ms  {100 * d['frac_of_8TBps_call']:5.1f} % of 8 TB/s per call ({d['algorithmic_read_bytes_per_row']:.3f} B/row)  redo {d['last_redo_ppm']} ppm\")
"
done
