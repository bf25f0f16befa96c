#!/usr/bin/env python3
"""Per-instance counter values of the k_adagrad_runs dispatches out of rocprofv3's JSON output (the CSV sums the instances).
   python3 tools/r03/channel_summary.py gpurun_out/r03/chan  ->  one JSON object per pass on stdout"""
import glob
import json
import os
import sys


def load(path):
    d = json.load(open(path))
    tool = d["rocprofiler-sdk-tool"]
    return tool[0] if isinstance(tool, list) else tool


def kernel_names(t):
    names = {}
    for k in t.get("kernel_symbols", []):
        names[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name") or ""
    return names


def counter_names(t):
    out = {}
    for a in t.get("counters", []):
        cid = a.get("id", {})
        out[cid.get("handle") if isinstance(cid, dict) else cid] = a.get("name")
    return out


def main(src):
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for p in glob.glob(os.path.join(d, "**", "*_results.json"), recursive=True):
            t = load(p)
            kn, cn = kernel_names(t), counter_names(t)
            recs = t.get("callback_records", {}).get("counter_collection", [])
            out = []
            for r in recs:
                info = r.get("dispatch_data", {}).get("dispatch_info", {})
                name = kn.get(info.get("kernel_id"), "")
                if "k_adagrad_runs" not in name:
                    continue
                per = {}
                for rec in r.get("records", []):
                    cid = rec.get("counter_id", {})
                    h = cid.get("handle") if isinstance(cid, dict) else cid
                    per.setdefault(cn.get(h, str(h)), []).append(rec.get("value"))
                out.append({"dispatch": info.get("dispatch_id"), "counters": per})
            print(json.dumps({"pass": os.path.basename(d), "file": os.path.relpath(p, src), "top_keys": sorted(t.keys()),
                              "n_dispatches": len(out), "dispatches": out}))


if __name__ == "__main__":
    main(sys.argv[1])
