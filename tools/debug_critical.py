"""Critical-path trace of the resident schedule from per-task time stamps (library built with -DAVBD_DF_STATS)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "Stress1000"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 420
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
g.step(1); g.w.synchronize()
IT = 20
buf = np.zeros((IT, 1024, 8), np.uint32)
g.pkg.lib().avbd_debug_task_times(g.w.h, buf.ctypes.data_as(C.c_void_p), IT)
forces = g.forces(); bodies = g.bodies()
dyn = bodies["invMass"] > 0
# wave -> body
w2b = buf[0, :, 5].astype(int); nW = int((buf[0, :, 0] > 0).sum())
b2w = {int(w2b[w]): w for w in range(1024) if buf[0, w, 0] > 0}
print("waves", nW)
t = buf.astype(np.int64)
T0 = t[0, :nW, 0].min()
def tm(it, b, k): return (t[it, b2w[b], k] - T0) / 100.0   # us
nbr = {b: set() for b in b2w}
for f in forces:
    a, b = int(f["bodyA"]), int(f["bodyB"])
    if int(f["numContacts"]) > 0 and a in b2w and b in b2w:
        nbr[a].add(b); nbr[b].add(a)
# find the last finishing task
end = max(((tm(IT - 1, b, 4), b) for b in b2w))
print("kernel span us", end[0])
it, b = IT - 1, end[1]
path = []
while True:
    deps = []
    for o in nbr[b]:
        if o > b: deps.append((tm(it, o, 3), o, it, "pose"))
        elif it > 0:
            deps.append((tm(it - 1, o, 4), o, it - 1, "dual"))
    if it > 0: deps.append((tm(it - 1, b, 4), b, it - 1, "self"))
    if not deps: break
    d = max(deps)
    path.append(dict(it=it, b=b, colo=int(b2w[b] // 4 == b2w[d[1]] // 4), dep=d[3], t_dep=d[0], t_start=tm(it, b, 0), t_pre=tm(it, b, 1), t_det=tm(it, b, 2), t_pub=tm(it, b, 3), t_dual=tm(it, b, 4), spins=int(t[it, b2w[b], 6]), nn=len(nbr[b])))
    b, it = d[1], d[2]
path.reverse()
print("hops", len(path))
import collections
agg = collections.defaultdict(list)
for p in path:
    ready = max(p["t_dep"], p["t_pre"])
    agg["dep->detect (dep after pre)"].append(p["t_det"] - p["t_dep"] if p["t_dep"] >= p["t_pre"] else np.nan)
    agg["pre-bound: pre_end - dep"].append(p["t_pre"] - p["t_dep"] if p["t_dep"] < p["t_pre"] else np.nan)
    agg["detect->publish"].append(p["t_pub"] - p["t_det"])
    agg["publish->dual"].append(p["t_dual"] - p["t_pub"])
    agg["hop (dep->this out)"].append((p["t_pub"] if True else 0) - p["t_dep"])
for k, v in agg.items():
    v = np.array(v, float); ok = ~np.isnan(v)
    print(f"{k:32s} n={ok.sum():4d} mean {np.nanmean(v):6.2f} p50 {np.nanmedian(v):6.2f} max {np.nanmax(v):6.2f} sum {np.nansum(v):8.1f}")
# same-XCD vs cross-XCD hand-offs (workgroups are dealt round-robin over the 8 XCDs; wave index = 4 * workgroup + wave)
allh = {"same": [], "cross": [], "workgroup": []}
for it in range(IT):
    for b in b2w:
        for o in nbr[b]:
            if o > b:
                d = tm(it, b, 2) - tm(it, o, 3)
                if 0 < d < 3.0 and int(t[it, b2w[b], 6]) >= 2:
                    allh["workgroup" if b2w[b] // 4 == b2w[o] // 4 else ("same" if (b2w[b] // 4) % 8 == (b2w[o] // 4) % 8 else "cross")].append(d)
for k, v in allh.items():
    if v: print(f"pose hand-offs that ended a real wait, {k}-XCD: n={len(v)} mean {np.mean(v):.3f} p50 {np.median(v):.3f} us")
kinds = collections.Counter((p["dep"], p["colo"]) for p in path)
print("dep kinds", kinds)
for p in path[:40]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items()})
