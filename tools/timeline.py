"""The kernels of the LAST call of a rocprofv3 --kernel-trace run of tools/shape_run.py, on one time axis (us from the start of the
call's fused pass): what runs back to back and where the stream idles.   python3 tools/timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
passes = [i for i, r in enumerate(rows) if "fused_" in r["Kernel_Name"] and "redo" not in r["Kernel_Name"]]
k, prev = passes[-1], passes[-2]
t0 = int(rows[k]["Start_Timestamp"])
print(f"previous call's pass started {(int(rows[prev]['Start_Timestamp']) - t0) / 1e3:.1f} us before this one")
last_end = None
for r in rows[prev + 1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if last_end is None else f"  (idle {(s - last_end) / 1e3:5.1f})"
    print(f"{r['Kernel_Name'][:60]:60s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f}{gap}")
    last_end = e
