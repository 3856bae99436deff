import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "Stress1000"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 420
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
import time
t0=time.perf_counter(); g.step(20); g.w.synchronize(); dt=(time.perf_counter()-t0)/20
st = g.w.stats()
print("ms/step", dt*1e3, "stats", st)
W = 8192
buf = np.zeros((W, 16), np.uint64)
g.pkg.lib().avbd_debug_iterate_stats(g.w.h, buf.ctypes.data_as(C.c_void_p), W)
act = buf[buf[:,5] > 0]
print("active waves", len(act), "tasks/wave mean", act[:,5].mean())
names = ["wait", "accumulate", "solve+store", "drain+publish", "dual+publish", "ntasks", "total", "loop-overhead", "mutable loads", "bodylocal+init", "eval+stores"]
tot = act[:,6].astype(np.float64)/100.0  # us
print("wave total us: mean %.1f max %.1f" % (tot.mean(), tot.max()))
print("shader clock GHz (s_memtime/s_memrealtime):", (act[:,11].astype(np.float64)/act[:,6]).mean()*0.1)
for i in (7,0,8,9,10,1,2,3,4):
    per_task = act[:,i].astype(np.float64)/100.0/act[:,5]
    print(f"  {names[i]:16s} per-task us mean {per_task.mean():8.2f} p50 {np.median(per_task):8.2f} max {per_task.max():8.2f}  share {act[:,i].sum()/act[:,6].sum():.3f}")
