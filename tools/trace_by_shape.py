#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid size): the bench runs ONE kernel template at several
shapes (headline 10k x 1M, north star 1k x 1M, C2, IVF coarse stage ...), which the plain --stats summary
averages together.  usage: trace_by_shape.py KERNEL_TRACE.csv OUT.csv"""
import csv, sys
from collections import defaultdict

LABELS = {}   # filled by hand in the committed copy's last column where it matters (see profiles/README)


def main():
    src, dst = sys.argv[1:3]
    acc = defaultdict(list)
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"]
        if "mrag::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        acc[(short, wgs, r["VGPR_Count"], r["LDS_Block_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    split = {}
    for key, d in acc.items():            # one grid can still serve several shapes (IVF coarse stage vs headline):
        d.sort()                          # split a group wherever consecutive sorted durations jump by > 2x
        cur = [d[0]]
        parts = [cur]
        for x in d[1:]:
            if x > 2 * cur[-1]:
                cur = []
                parts.append(cur)
            cur.append(x)
        for i, part in enumerate(parts):
            split[key + (i,)] = part
    for (short, wgs, vgpr, lds, _), d in split.items():
        label = next((v for (s, w), v in LABELS.items() if s in short and w == wgs), "")
        rows.append((sum(d), short, wgs, vgpr, lds, len(d), sum(d) / len(d) / 1e3, d[0] / 1e3, d[len(d) // 2] / 1e3, d[-1] / 1e3, label))
    rows.sort(reverse=True)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroups", "vgpr", "lds_bytes", "calls", "avg_us", "min_us", "median_us", "max_us", "bench_leg"])
        for r in rows:
            w.writerow([r[1], r[2], r[3], r[4], r[5], f"{r[6]:.1f}", f"{r[7]:.1f}", f"{r[8]:.1f}", f"{r[9]:.1f}", r[10]])
    for r in rows[:14]:
        print(f"{r[1][:60]:60s} wgs={r[2]:6d} calls={r[5]:4d} avg={r[6]:10.1f} us min={r[7]:10.1f} med={r[8]:10.1f}")


if __name__ == "__main__":
    main()
