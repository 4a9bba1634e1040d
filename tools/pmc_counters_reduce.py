#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (tools/profile_counters.sh) to one row per (kernel, launch shape): counter values per
launch, derived MFMA-busy / wait shares, L2 hit rate, fabric traffic per launch (gfx950 FETCH_SIZE x2 correction,
MI355X_MICROARCH.md HBM section).  usage: pmc_counters_reduce.py OUT.json PASS_DIR..."""
import csv, glob, json, sys
from collections import defaultdict

N_CU, SIMD_PER_CU, N_XCD = 256, 4, 8


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.replace("mrag::", "")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    # (kernel, grid, workgroup) -> counter -> [values per dispatch]
    acc = defaultdict(lambda: defaultdict(dict))
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "mrag::" not in r["Kernel_Name"]:
                    continue
                key = (short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0))
                c = acc[key][r["Counter_Name"]]
                c[r["Dispatch_Id"]] = c.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    rows = []
    for (k, grid, wg), cs in acc.items():
        # one (kernel, grid) can still serve two shapes (C2 and the north star both launch 256 workgroups): split the
        # dispatches into clusters by GRBM_GUI_ACTIVE / SQ_WAVE_CYCLES / FETCH_SIZE magnitude (> 2.5x apart)
        base = next((cs[n] for n in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "TCC_REQ_sum") if n in cs), None)
        e = {"kernel": k, "workgroups": grid // max(1, wg), "launches": max(len(v) for v in cs.values())}
        for name, per_disp in cs.items():
            vals = sorted(per_disp.values())
            # keep the heaviest cluster (the largest shape this kernel runs at this grid)
            top = [v for v in vals if v * 2.5 >= vals[-1]]
            e[name] = sum(top) / len(top)
            e[name + "_launches"] = len(top)
        g = e.get("GRBM_GUI_ACTIVE")
        if g:
            cyc = g / N_XCD                                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            e["gpu_cycles_per_launch"] = cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy_frac_of_all_simds"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_CU * SIMD_PER_CU)
            if "SQ_BUSY_CYCLES" in e:
                e["sq_busy_frac"] = e["SQ_BUSY_CYCLES"] / (cyc * N_XCD) if e["SQ_BUSY_CYCLES"] else None
        w = e.get("SQ_WAVE_CYCLES")
        if w:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in e:
                    e[n.lower() + "_share_of_wave_cycles"] = e[n] / w
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        if "FETCH_SIZE" in e:
            e["fabric_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["traffic_bytes_per_launch"] = e.get("fabric_read_bytes_corrected", 0.0) + e.get("write_bytes", 0.0)
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        rows.append(e)
    rows.sort(key=lambda e: -(e.get("SQ_WAVE_CYCLES") or e.get("FETCH_SIZE") or 0))
    json.dump({"note": "per launch; for a (kernel, grid) that runs at several shapes the heaviest cluster of dispatches; FETCH_SIZE "
                       "doubled (gfx950), L2<->fabric requests (Infinity-Cache hits included); SQ_* cycle counters are in quad-cycles "
                       "except SQ_VALU_MFMA_BUSY_CYCLES (cycles); GRBM_GUI_ACTIVE summed over 8 XCDs",
               "kernels": rows}, open(out, "w"), indent=1)
    for e in rows[:24]:
        print(f"{e['kernel'][:52]:52s} wgs={e['workgroups']:6d} "
              f"mfma_busy={e.get('mfma_busy_frac_of_all_simds', float('nan')):.3f} wait={e.get('sq_wait_any_share_of_wave_cycles', float('nan')):.3f} "
              f"l2hit={e.get('l2_hit_rate', float('nan')):.3f} traffic={e.get('traffic_bytes_per_launch', float('nan')) / 1e9:.3f} GB")


if __name__ == "__main__":
    main()
