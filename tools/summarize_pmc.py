"""Summarise rocprofv3 --pmc counter_collection.csv files into one JSON (mean per dispatch and kernel).
Usage: python tools/summarize_pmc.py out.json dir_or_csv [dir_or_csv ...]"""
import collections, csv, glob, json, os, re, sys

def short(name):
    m = re.search(r"(pair_accumulate|plan_tiles|propose_lattice|field_update|propose|apply|claim|field_sites)", name)
    return m.group(1) if m else name.split("(")[0][:40]

out = collections.defaultdict(dict)
for arg in sys.argv[2:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            v = v[len(v) // 5:]                       # drop warm-up dispatches
            key = c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c
            out[k][key] = sum(v) / len(v)
json.dump({"note": "rocprofv3 --pmc, separate passes per counter group, mean per dispatch over the timed steps of "
                   "`bench.py --no-cpu-baseline` (BASELINE config 2). FETCH_SIZE/WRITE_SIZE in KB as reported; per "
                   "MI355X_MICROARCH.md gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x "
                   "(these kernels issue 4-byte-per-lane loads: uncalibrated).",
           "per_dispatch": out}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in ("pair_accumulate", "field_update", "propose_lattice", "apply") if k in out}, indent=1))
