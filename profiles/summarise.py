#!/usr/bin/env python3
"""Reduce the rocprofv3 output of profiles/collect.sh to the small summaries kept in profiles/.

    python3 profiles/summarise.py gpurun_out/prof_<tag> <tag>

Writes (into gpurun_out/prof_<tag>/summary/, to be copied to profiles/):
  <tag>_kernel_stats.csv                 rocprofv3's own --stats table of the kernel-trace pass
  <tag>_pmc_FETCH_SIZE_by_kernel.csv     per-kernel mean of FETCH_SIZE (KB as reported; x2 on gfx950, MI355X_MICROARCH.md)
  <tag>_pmc_WRITE_SIZE_by_kernel.csv     per-kernel mean of WRITE_SIZE (KB)
  <tag>_window_summary.json              dominant kernel in the steady window only (the timed + profiled region)
  <tag>_bench.json, <tag>_bench_under_rocprof.json
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(src, "summary")
os.makedirs(out, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[-1] if f else None


def short(name):
    """'void avbd::k_iterate<1, false>(avbd::BodyTable, ...)' -> 'k_iterate<1, false>'"""
    head = name.split("(")[0]
    return head.split("::")[-1] if "::" in head else head.split(" ")[-1]


summary = {}
stats = one("trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
trace = one("trace/**/*kernel_trace.csv")
if trace:
    rows = [r for r in csv.DictReader(open(trace))]
    per = defaultdict(list)
    for r in rows:
        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    dom = max(per, key=lambda k: sum(e - s for s, e in per[k]))
    d = sorted(per[dom])
    dur = [(e - s) / 1e3 for s, e in d]
    # bench.py --steps 100 --warmup 10: the last 150 launches are the timed (100) + HIP-event-profiled (50) steady window
    summary.update({"dominant_kernel": dom, "dispatches": len(dur), "avg_us_all": sum(dur) / len(dur),
                    "avg_us_last150_steady_window": sum(dur[-150:]) / len(dur[-150:])})
    # per-kernel mean inside the same window (by time stamp)
    t0 = d[-150][0] if len(d) >= 150 else d[0][0]
    win = {}
    for k, v in per.items():
        w = [(e - s) / 1e3 for s, e in v if s >= t0]
        if w:
            win[k] = {"calls_per_step": round(len(w) / min(150, len(d)), 3), "avg_us": round(sum(w) / len(w), 2),
                      "us_per_step": round(sum(w) / min(150, len(d)), 2)}
    summary["steady_window_by_kernel"] = dict(sorted(win.items(), key=lambda kv: -kv[1]["us_per_step"]))

for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(("fetch" if ctr == "FETCH_SIZE" else "write") + "/**/*counter_collection.csv")
    if not f:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == ctr:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_{ctr}_by_kernel.csv"), "w") as o:
        o.write("kernel,dispatches,mean_%s_KB,mean_last40_KB\n" % ctr)
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            o.write("%s,%d,%.3f,%.3f\n" % (k, len(v), sum(v) / len(v), sum(v[-40:]) / len(v[-40:])))
    dom = summary.get("dominant_kernel")
    if dom in per:
        # bench.py --steps 20 --warmup 5: last 40 launches = timed (20) + profiled (20) steady window
        summary[f"{ctr}_KB_last40"] = sum(per[dom][-40:]) / len(per[dom][-40:])

json.dump(summary, open(os.path.join(out, f"{tag}_window_summary.json"), "w"), indent=1)
for a, b in (("bench.json", f"{tag}_bench.json"), ("bench_trace.json", f"{tag}_bench_under_rocprof.json")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(out, b), "w").write(lines[-1])
print(json.dumps(summary, indent=1))
