#!/usr/bin/env python3
"""Reduce the rocprofv3 output of profiles/collect_cli.sh to one JSON + the kernel stats table.

    python3 profiles/summarise_cli.py gpurun_out/prof_<tag> <tag> <profiled steps>

Writes gpurun_out/prof_<tag>/summary/<tag>_cli_summary.json (per kernel: calls per step, mean us; per counter: mean per dispatch
of the dominant kernel, bytes for FETCH_SIZE / WRITE_SIZE as reported by rocprofv3, i.e. KB -> see MI355X_MICROARCH.md for the
gfx950 factor 2 on FETCH_SIZE) and <tag>_kernel_stats.csv."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = os.path.join(src, "summary"); os.makedirs(out, exist_ok=True)

def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[-1] if f else None

def short(name):
    head = name.split("(")[0]
    return head.split("::")[-1] if "::" in head else head.split(" ")[-1]

S = {"tag": tag}
try:   # what the profile was collected on: sha256 over the library's sources (tools/source_id.py) and the size of the binary that ran
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import source_id as SID
    S["source_id"] = SID.source_id()
    S["library_bytes"] = os.path.getsize(os.path.join(SID.ROOT, "avbd-demo3d_amd", "libavbd_hip.so"))
    S["cli_flags"] = os.environ.get("AVBD_CLI_EXTRA", "")
except Exception as e:
    S["source_id"] = None
for n in ("presim", "plain", "trace"):
    p = os.path.join(src, n + ".json")
    if os.path.exists(p):
        try: S[n + "_run"] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception: S[n + "_run"] = open(p).read()[-300:]
stats = one("trace/**/*kernel_stats.csv")
if stats: shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
trace = one("trace/**/*kernel_trace.csv")
dom = None
if trace:
    per = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    dom = max(per, key=lambda k: sum(per[k]))
    S["dominant_kernel"] = dom
    S["by_kernel_50_steps"] = {k: {"calls_per_step": round(len(v) / 50.0, 3), "avg_us": round(sum(v) / len(v), 2), "us_per_step": round(sum(v) / 50.0, 2)}
                               for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))}
S["counters_per_dispatch_of_dominant_kernel"] = {}
for d in sorted(glob.glob(os.path.join(src, "*"))):
    f = one(os.path.basename(d) + "/**/*counter_collection.csv")
    if not f: continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if dom is None or k == dom:
            for c, v in cs.items():
                S["counters_per_dispatch_of_dominant_kernel"][c] = {"mean": sum(v) / len(v), "dispatches": len(v), "pass": os.path.basename(d)}
json.dump(S, open(os.path.join(out, f"{tag}_cli_summary.json"), "w"), indent=1)
print(json.dumps(S, indent=1)[:6000])
