import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "Stress1000"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 420
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
import time
t0=time.perf_counter(); g.step(20); g.w.synchronize(); dt=(time.perf_counter()-t0)/20
st = g.w.stats()
print("ms/step", dt*1e3, "schedule", st["schedule"])
W = 2048
buf = np.zeros((W, 16), np.uint64)
g.pkg.lib().avbd_debug_iterate_stats(g.w.h, buf.ctypes.data_as(C.c_void_p), W)
act = buf[buf[:,5] > 0].astype(np.float64)
print("active waves", len(act), "iterations", act[:,5].mean(), "entries mean", act[:,8].mean(), "max", act[:,8].max())
names = {9:"pick (out of order)", 10:"load task", 7:"pre-wait work", 0:"wait", 1:"eval-post", 2:"accumulate", 3:"solve+publish", 4:"dual", 11:"park task"}
print("wave total us: mean %.1f max %.1f" % ((act[:,6]/100).mean(), (act[:,6]/100).max()))
heavy = act[act[:,8] >= 8]
for label, sel in (("all", act), ("entries>=8", heavy)):
    print(label, len(sel))
    for i in (9,10,7,0,1,2,3,4,11):
        per = sel[:,i]/100.0/sel[:,5]
        print(f"  {names[i]:16s} per-task us mean {per.mean():8.2f} p50 {np.median(per):8.2f} max {per.max():8.2f}")
print("poll rounds per task", (act[:,14]/act[:,5]).mean())
w = act[act[:,13] > 0]
print("publish->detect latency us (real waits): %.2f over %d samples" % ((w[:,12].sum()/w[:,13].sum())/100.0, int(w[:,13].sum())))
