"""Collapse rocprofv3 counter-collection CSVs into a per-kernel mean table (what gets committed
under profiles/; the raw per-dispatch CSVs stay in gpurun_out/)."""
import collections
import csv
import sys


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches"] + counters)
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        w.writerow([k, n] + ["%.1f" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in counters])


if __name__ == "__main__":
    main(sys.argv[1:])
