#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs (one pass per counter) to bytes per launch per kernel.
usage: pmc_reduce.py OUT.json DIR_FETCH DIR_WRITE
FETCH_SIZE / WRITE_SIZE are reported in KB summed over dispatch; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys
from collections import defaultdict


def per_launch(d, counter):
    acc, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            acc[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / max(1, len(n[k])) for k in acc}, {k: len(v) for k, v in n.items()}


def short(k):
    for s in ("bf_gemm_topk", "bf_stream_topk", "bf_merge", "prep_rows"):
        if s in k:
            return s
    return None


def main():
    out, dfetch, dwrite = sys.argv[1:4]
    fe, nf = per_launch(dfetch, "FETCH_SIZE")
    wr, _ = per_launch(dwrite, "WRITE_SIZE")
    res = {}
    for k, v in fe.items():
        s = short(k)
        if not s:
            continue
        e = res.setdefault(s, {"launches_seen": 0, "FETCH_SIZE_KB_per_launch": 0.0, "WRITE_SIZE_KB_per_launch": 0.0})
        # several template instances may share a short name: keep the heaviest
        if v >= e["FETCH_SIZE_KB_per_launch"]:
            e.update(FETCH_SIZE_KB_per_launch=v, WRITE_SIZE_KB_per_launch=wr.get(k, 0.0), launches_seen=nf[k])
    for s, e in res.items():
        e["fabric_read_bytes_corrected"] = e["FETCH_SIZE_KB_per_launch"] * 1024 * 2
        e["write_bytes"] = e["WRITE_SIZE_KB_per_launch"] * 1024
        e["traffic_bytes_per_launch"] = e["fabric_read_bytes_corrected"] + e["write_bytes"]
    json.dump({"note": "FETCH_SIZE doubled per the gfx950 correction; L2<->fabric requests (Infinity-Cache hits included)",
               "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
