#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per-kernel, per-launch averages of every counter found under the given directories.

usage: tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write ... > profiles/rNN_pmc_summary.csv
Only launches whose grid covers the benchmark batch are of interest, so the tiny warm-up launches of a different grid size are
kept apart by reporting per (kernel, grid size) and sorting by total counter mass; pass --top to keep the N largest grids.
"""
import csv, glob, os, sys, collections, sqlite3

def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    key = (name, row.get("Grid_Size", ""))
                    a = acc[key][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
        for path in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):  # rocpd (sqlite) output
            db = sqlite3.connect(path)
            for name, grid, cname, val in db.execute("select kernel_name, grid_size, counter_name, value from counters_collection"):
                a = acc[(name.split("(")[0], str(grid))][cname]
                a[0] += float(val); a[1] += 1
    counters = sorted({c for v in acc.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid", "launches"] + counters)
    for (name, grid), v in sorted(acc.items()):
        n = max(x[1] for x in v.values())
        w.writerow([name, grid, n] + [("%.1f" % (v[c][0] / v[c][1]) if c in v else "") for c in counters])

if __name__ == "__main__":
    main()
