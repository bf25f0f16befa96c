#!/usr/bin/env python3
"""The exchange pass and the builder out of tools/r03/take_profiles.sh's rocprofv3 output -> summary/r03_turn_*.{json,csv}, summary/r03_bca_*."""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1]
out = os.path.join(src, "summary"); os.makedirs(out, exist_ok=True)

def stats(d, pick):
    rows = []
    for f in glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, "r03_%s_kernel_stats.csv" % d.replace("_kt", "")))
        for row in csv.DictReader(open(f)):
            if any(p in row["Name"] for p in pick):
                rows.append({"kernel": row["Name"][:110], "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6, "max_ms": float(row["MaxNs"]) / 1e6, "total_ms": float(row["TotalDurationNs"]) / 1e6})
    return rows

def counters(dirs, pick):
    acc = collections.defaultdict(list)
    for d in dirs:
        for c in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(c)):
                if any(p in row["Kernel_Name"] for p in pick):
                    acc[(row["Kernel_Name"][:90], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc

# ---- exchange pass
t = {"tool": "tools/r03/turn_bench.py 200 f32 (V = 5 M context rows, dim 200, bf16 wire, null transport)", "kernels": stats("turn_kt", ("k_sync_turn", "k_exchange_turn", "k_hub"))}
acc = counters(("turn_pmcA", "turn_pmcB"), ("k_sync_turn_flat4",))
per = collections.defaultdict(dict)
for (k, c), v in acc.items():
    per[k][c] = sum(v) / len(v)
t["counters_mean_per_launch"] = per
for k, c in per.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["traffic_bytes"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024          # FETCH_SIZE x2: gfx950 correction (MI355X_MICROARCH.md)
for f in sorted(glob.glob(os.path.join(src, "turn_b*.json"))):
    t.setdefault("turn_ms_by_blocks_per_cu", {})[os.path.basename(f)[6:-5]] = json.loads(open(f).read().strip().splitlines()[-1])["turn_ms"]
elems = 5_000_000 * 200
for row in t["kernels"]:
    if "flat4<true, true, true" in row["kernel"].replace("(bool)1", "true").replace("(bool)0", "false"):
        row["algorithmic_bytes"] = elems * 22; row["TBps"] = elems * 22 / (row["avg_ms"] * 1e-3) / 1e12; row["frac_of_8TBps"] = row["TBps"] / 8
json.dump(t, open(os.path.join(out, "r03_turn_summary.json"), "w"), indent=1)
print("turn:", [(r["kernel"][40:80], round(r["avg_ms"], 3)) for r in t["kernels"]][:6])

# ---- builder
b = {"tool": "tests/tools/bca_bench.py (DBLP-like graph, V = 300 050)", "kernels": stats("bca_kt", ("k_bca", "k_totals", "k_gather_rows"))}
acc = counters(("bca_pmcA", "bca_pmcB"), ("k_bca",))
tot = collections.defaultdict(float)
for (k, c), v in acc.items():
    tot[("lds" if "true" in k or "(bool)1" in k else "global") + ":" + c] += sum(v)
b["counters_summed_over_launches"] = tot
log = open(os.path.join(src, "bca_bench.log")).read()
b["log_tail"] = [l for l in log.splitlines() if "ge_bca_build" in l or l.startswith("device") or l.startswith("oracle")][-9:]
json.dump(b, open(os.path.join(out, "r03_bca_summary.json"), "w"), indent=1)
print("bca:", [(r["kernel"][20:60], r["calls"], round(r["total_ms"], 2)) for r in b["kernels"]])
