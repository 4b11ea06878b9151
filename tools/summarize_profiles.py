#!/usr/bin/env python3
"""Condense rocprofv3 output directories (see collect_profiles.sh) into small summaries fit for profiles/."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
os.makedirs(os.path.join(out, "summary"), exist_ok=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.strip()


# ---- kernel stats (rocprofv3's own summary) and per-kernel averages from the trace
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "summary", "kernel_stats.csv"), "w") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        r["Percentage"], "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])
    print("kernel_stats.csv <-", f)

# ---- counters: average per dispatch, per kernel
pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            pmc[k][c][0] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], c)
            if key not in seen:
                seen.add(key)
                pmc[k][c][1] += 1
res = {}
for k, cs in pmc.items():
    res[k] = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    res[k]["dispatches"] = max(v[1] for v in cs.values())
    if "FETCH_SIZE" in res[k] or "WRITE_SIZE" in res[k]:
        # MI355X_MICROARCH.md "HBM": both counters are in KiB... units of 1 KB; FETCH_SIZE reports half the bytes of wide
        # coalesced reads on gfx950 -> doubled; WRITE_SIZE exact
        res[k]["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * res[k].get("FETCH_SIZE", 0.0)
        res[k]["hbm_write_bytes_per_launch"] = 1024.0 * res[k].get("WRITE_SIZE", 0.0)
        res[k]["hbm_bytes_per_launch"] = res[k]["hbm_read_bytes_per_launch"] + res[k]["hbm_write_bytes_per_launch"]
# which workload the counters belong to (bench.py checks this before quoting a traffic figure)
for f in glob.glob(os.path.join(out, "pmc_FETCH_SIZE.json")):
    try:
        line = [l for l in open(f) if l.startswith("{")][-1]
        cfg = json.loads(line)["config"]
        ncalls = json.loads(line).get("fwd_bwd_calls")         # eager executions of the headline step in a counter pass (--headline-only --no-graph)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        res["_meta"] = {"workload": cfg["workload"].split("; fwd")[0], "train": cfg["dropout"].startswith("train"),
                        "csrc_sha": bench.kernel_source_sha(), "steps_profiled": ncalls,
                        "command": "rocprofv3 --kernel-trace --pmc <one counter group per run> -- python3 bench.py --steps 3 --warmup 1 "
                                   "--profile-steps 0 --no-graph --headline-only --no-cpu-baseline",
                        "units": "averages per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_* bytes = 2*1024*FETCH_SIZE + "
                                 "1024*WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md 'HBM'); SQ_* cycle counters in quad-cycles "
                                 "except SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES"}
    except (IndexError, KeyError, ValueError) as e:
        print("no _meta:", e)
json.dump(res, open(os.path.join(out, "summary", "pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
print("pmc_per_kernel.json: %d kernels" % len(res))
for k in sorted((k for k in res if k != "_meta"), key=lambda k: -res[k].get("SQ_BUSY_CYCLES", res[k].get("hbm_bytes_per_launch", 0))):
    r = res[k]
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    print("%-44s n=%-4d" % (k[:44], r["dispatches"]), " ".join("%s=%.4g" % (c, v) for c, v in sorted(r.items()) if c != "dispatches"))
