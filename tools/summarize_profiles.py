"""Boil the rocprofv3 output of tools/profile_round.sh down to the small files kept under profiles/:
  <tag>_<workload>_bench.json          the bench / shape line of the same command
  <tag>_<workload>_kernel_stats.csv    --kernel-trace --stats (every kernel of the run)
  <tag>_<workload>_pmc_fetch_size.csv / _pmc_write_size.csv   PMC rows of the workload's dominant kernels
  <tag>_traffic.json                   HBM bytes per launch per workload, corrected as MI355X_MICROARCH.md (HBM section)
                                       prescribes: FETCH_SIZE x 1024 (KiB) x 2 (gfx950 tallies the 128-B requests of a
                                       16 B/lane streaming read at 64 B), WRITE_SIZE x 1024
"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
DOMINANT = {"filter_agg": ["filter_agg_kernel"], "bool_xb": ["fused_filter_compact", "bits_compact_kernel"], "bool_x": ["fused_filter_compact", "mask_select", "compact_ranges", "scan_"],
            "strings": ["fused_filter_compact", "str_gather", "sel_", "scan_", "str_sums", "str_group"],
            "strings_dense": ["fused_direct_compact", "sel_str_tile", "str_sums", "str_group"],
            "dense1": ["fused_direct_compact"], "dense3": ["fused_direct_compact"],
            "wide5": ["fused_filter_compact", "fused_direct_compact", "compact_ranges"], "wide9": ["fused_filter_compact", "fused_direct_compact", "compact_ranges"],
            "wide9n": ["fused_filter_compact", "fused_direct_compact", "compact_ranges", "bits_compact"], "wide9n_dense": ["fused_filter_compact", "fused_direct_compact", "compact_ranges", "bits_compact"],
            "wide5_dense": ["fused_filter_compact", "fused_direct_compact"], "wide9_dense": ["fused_filter_compact", "fused_direct_compact"]}  # (their first, unprepared call: fused_filter_compact + fused_redo_tiles)


def find(base, sub, suffix):
    hits = sorted(glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


traffic = {}
for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    w = os.path.basename(wdir)
    if w == "summary" or not os.path.isdir(wdir):
        continue
    if os.path.exists(os.path.join(wdir, "bench.json")):
        shutil.copy(os.path.join(wdir, "bench.json"), os.path.join(dst, f"{tag}_{w}_bench.json"))
    stats = find(wdir, "stats", "kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))
        for r in list(csv.reader(open(stats)))[1:4]:
            print(w, "stats:", r[0][:70], r[1:4])
    names = DOMINANT.get(w, ["fused_filter_compact"])
    if w.rstrip("0123456789km") in ("iid", "sorted", "sorteddesc", "clustered"):  # tools/shape_run.py: tables that are not independent rows
        names = ["fused_filter_compact", "fused_direct_compact", "fused_redo_waves"]
    entry = {}
    for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = find(wdir, sub, "counter_collection.csv")
        if not path:
            print(w, "missing", sub)
            continue
        keep, per_kernel = [], {}
        for r in csv.DictReader(open(path)):
            kn = r.get("Kernel_Name", "")
            if r.get("Counter_Name") == counter and any(nm in kn for nm in names):
                keep.append(r)
                per_kernel.setdefault(kn.split("(")[0][:90], []).append(float(r["Counter_Value"]))
        if keep:
            with open(os.path.join(dst, f"{tag}_{w}_pmc_{counter.lower()}.csv"), "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
                wr.writeheader()
                wr.writerows(keep)
            entry[counter] = {k: sum(v) / len(v) for k, v in per_kernel.items()}
    if entry:
        # per query: the sum over the dominant kernels of (average per launch x launches per query); one launch each here
        rd = sum(entry.get("FETCH_SIZE", {}).values()) * 1024 * 2
        wrb = sum(entry.get("WRITE_SIZE", {}).values()) * 1024
        traffic[w] = {"fetch_size_raw_avg_per_kernel": entry.get("FETCH_SIZE"), "write_size_raw_avg_per_kernel": entry.get("WRITE_SIZE"),
                      "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wrb, "hbm_bytes_per_launch": rd + wrb}
        print(w, "traffic GB read/write:", rd / 1e9, wrb / 1e9)
traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); FETCH_SIZE x 1024 x 2 "
                    "(gfx950 counts 128-B requests of a 16 B/lane stream at 64 B), WRITE_SIZE x 1024; average per launch of the dominant kernels")
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
