import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
g = H.HipWorld(); g.scene(sys.argv[1] if len(sys.argv) > 1 else "Stress1000"); g.step(420); g.w.synchronize()
acc = np.zeros(6)
for k in range(20):
    g.step(1); g.w.synchronize()
    b = np.zeros(7, np.uint64)
    rc = g.pkg.lib().avbd_debug_topology_times(g.w.h, b.ctypes.data_as(C.c_void_p))
    d = np.diff(b.astype(np.int64)) / 100.0
    d[(d < 0) | (d > 1e6)] = 0.0          # a phase the kernel skipped leaves no stamp (the default order needs no levels)
    acc += d
names = ["predict+flags", "scan", "scatter", "rank+expand", "levels | order by index", "order | place_chains"]
for n, v in zip(names, acc / 20): print(f"{n:16s} {v:6.2f} us")
print("total", acc.sum() / 20)
