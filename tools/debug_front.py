import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
g = H.HipWorld(); g.scene(sys.argv[1] if len(sys.argv) > 1 else "Stress1000"); g.step(420); g.w.synchronize()
acc = np.zeros(3)
for k in range(20):
    g.step(1); g.w.synchronize()
    b = np.zeros(4, np.uint64)
    g.pkg.lib().avbd_debug_front_times(g.w.h, b.ctypes.data_as(C.c_void_p))
    acc += np.diff(b.astype(np.int64)) / 100.0
for n, v in zip(["loads", "collide+initialize", "decay+append+stores"], acc / 20): print(f"{n:22s} {v:6.2f} us")
