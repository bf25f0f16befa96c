"""Turns the rocprofv3 output of tools/r03/take_profiles.sh into the files committed under profiles/:
   r03_bench.json                     the un-profiled bench line (default layout)
   r03_bench_kernel_stats.csv         rocprofv3 --kernel-trace --stats
   r03_bench_pmc_rows.csv             the counter rows of the update kernel (all passes)
   r03_bench_pmc_summary.json         means per launch, derived HBM traffic (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)
   r03_layout_evidence.json           the same summary for --layout separate_tables next to the default (counter evidence of the record layout)
   traffic.json                       what bench.py reads for roofline.traffic (keyed by kernel instance, workload, layout, schedule bytes)
usage: python3 tools/r03/profile_summary.py gpurun_out/r03/prof"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_dir = os.path.join(src, "summary")
os.makedirs(out_dir, exist_ok=True)


def bench_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no bench line in " + path)


def counters(layout):
    acc = collections.defaultdict(list)
    rows = []
    for d in sorted(glob.glob(os.path.join(src, "pmc?_" + layout))):
        for c in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(c)):
                if "k_adagrad_runs" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    rows.append(row)
    return acc, rows


def kernel_stats(layout):
    for c in glob.glob(os.path.join(src, "kt_" + layout, "**", "*kernel_stats.csv"), recursive=True):
        return c
    return None


summary = {}
traffic_entries = []
for layout in ("default", "dim100", "bf16d300"):
    if not os.path.exists(os.path.join(src, "bench_%s.json" % layout)):
        continue
    b = bench_line(os.path.join(src, "bench_%s.json" % layout))
    acc, rows = counters(layout)
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    s = {"layout": layout, "bench": {k: b[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "roofline": b["roofline"], "trainer": b["trainer"],
         "counters_mean_per_launch": mean, "launches": {k: len(v) for k, v in acc.items()}}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch_raw, write = mean["FETCH_SIZE"] * 1024, mean["WRITE_SIZE"] * 1024          # the counters are in KiB
        s["derived"] = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2 * fetch_raw, "write_bytes": write,
                        "traffic_bytes_per_launch": 2 * fetch_raw + write,
                        "traffic_over_schedule_bytes": (2 * fetch_raw + write) / b["roofline"]["schedule_bytes_per_launch"],
                        "traffic_TBps_at_unprofiled_kernel_ms": (2 * fetch_raw + write) / (b["roofline"]["kernel_ms"] * 1e-3) / 1e12}
        if "TCC_HIT_sum" in mean:
            s["derived"]["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in mean:
        s.setdefault("derived", {})["ea_read_latency_units"] = mean["TCC_EA0_RDREQ_LEVEL_sum"] / mean["TCC_EA0_RDREQ_sum"]
        s["derived"]["ea_write_latency_units"] = mean["TCC_EA0_WRREQ_LEVEL_sum"] / mean["TCC_EA0_WRREQ_sum"]
    # kernel times of the profiled runs (HIP events inside bench.py), to show which box / mode each pass ran in
    s["kernel_ms_by_pass"] = {}
    for p in ("kt", "pmcA", "pmcB", "pmcC", "pmcD"):
        f = os.path.join(src, "bench_%s_%s.json" % (p, layout))
        if os.path.exists(f):
            s["kernel_ms_by_pass"][p] = bench_line(f)["roofline"]["kernel_ms"]
    summary[layout] = s
    if layout == "default":
        json.dump(b, open(os.path.join(out_dir, "r03_bench.json"), "w"), indent=1)
        ks = kernel_stats(layout)
        if ks:
            shutil.copy(ks, os.path.join(out_dir, "r03_bench_kernel_stats.csv"))
        if rows:
            with open(os.path.join(out_dir, "r03_bench_pmc_rows.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        json.dump(s, open(os.path.join(out_dir, "r03_bench_pmc_summary.json"), "w"), indent=1)
    if layout in ("dim100", "bf16d300"):
        json.dump(s, open(os.path.join(out_dir, "r03_%s_pmc_summary.json" % layout), "w"), indent=1)
        ks = kernel_stats(layout)
        if ks:
            shutil.copy(ks, os.path.join(out_dir, "r03_%s_kernel_stats.csv" % layout))
    if layout in ("default", "dim100", "bf16d300") and "derived" in s and "traffic_bytes_per_launch" in s["derived"]:
        c = b["config"]
        traffic_entries.append({"kernel": b["roofline"]["kernel"], "vocab": c["vocab"], "nnz_per_gpu": c["nnz_per_gpu"], "dim": c["dim"], "cost": c["cost"],
                                "layout": "", "schedule_bytes": b["roofline"]["schedule_bytes_per_launch"], "row_stride": b["trainer"]["row_stride"],
                                "traffic_bytes_per_launch": s["derived"]["traffic_bytes_per_launch"],
                                "source": "profiles/r03_%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, FETCH_SIZE x2 per MI355X_MICROARCH.md)"
                                          % ("bench" if layout == "default" else layout)})
if traffic_entries:
    json.dump(traffic_entries, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
for layout, s in summary.items():
    d = s.get("derived", {})
    print(layout, "kernel_ms", round(s["roofline"]["kernel_ms"], 2), "frac", round(s["roofline"]["frac"], 3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}, s["kernel_ms_by_pass"])
