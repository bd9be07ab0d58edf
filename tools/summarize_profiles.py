"""Boil the rocprofv3 output of tools/profile_round.sh down to the small files kept under profiles/."""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, suffix):
    hits = sorted(glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


stats = find("stats", "kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    for r in rows[:6]:
        print("stats:", r[:8])

traffic = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = find(sub, "counter_collection.csv")
    if not path:
        print("missing", sub)
        continue
    rd = csv.DictReader(open(path))
    keep, vals = [], []
    for r in rd:
        if "fused_filter_compact" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
            keep.append(r)
            vals.append(float(r["Counter_Value"]))
    if keep:
        with open(os.path.join(dst, f"{tag}_pmc_{counter.lower()}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)
        traffic[counter] = sum(vals) / len(vals)
        print(counter, "launches", len(vals), "avg", traffic[counter], "kernel", keep[0]["Kernel_Name"][:80])
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_raw.json"), "w"), indent=1)
