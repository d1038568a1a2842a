"""Summarise one rocprofv3 PMC pass of SQ counters (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE, with --kernel-trace, csv output) into per-kernel averages and ratios:
mfma_busy_over_busy_cu (of 4 SIMDs), wait_any_over_wave_cycles, lds_conflict_over_active.
Usage: python tools/pmc_sq_summary.py <pass_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    f = (glob.glob(d + "/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        row = {n: sum(v) / len(v) for n, v in c.items()}
        row["launches"] = max(len(v) for v in c.values())
        if row.get("SQ_BUSY_CU_CYCLES"):
            row["mfma_busy_over_busy_cu"] = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / row["SQ_BUSY_CU_CYCLES"]
        if row.get("SQ_WAVE_CYCLES"):
            row["wait_any_over_wave_cycles"] = row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"]
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_over_active"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
        out[k] = row
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0) * kv[1]["launches"])[:12]:
        print("%-64s launches %5d  mfma/4simd %.3f  wait_any %.3f  lds_conflict %.4f" % (
            k, v["launches"], v.get("mfma_busy_over_busy_cu", 0) / 4, v.get("wait_any_over_wave_cycles", 0), v.get("lds_conflict_over_active", 0)))


if __name__ == "__main__":
    main()
