"""Summarise two rocprofv3 PMC passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, each with --kernel-trace, csv output)
into profiles/pmc_hbm_traffic.json: per kernel x grid size, average KB fetched/written and corrected HBM bytes.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies wide coalesced reads at half their bytes -> x2;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def load(d, cname):
    f = (glob.glob(d + "/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv"))[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            out[(r["Kernel_Name"].split("(")[0][:48], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return out


def main():
    F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(F, key=lambda k: -sum(F[k])):
        f = sum(F[k]) / len(F[k])
        w = sum(W.get(k, [0])) / max(len(W.get(k, [0])), 1)
        res["%s|%s" % k] = dict(launches=len(F[k]), fetch_size_kb=f, write_size_kb=w, hbm_bytes_corrected=(2 * f + w) * 1024)
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(k, v)


if __name__ == "__main__":
    main()
