"""How many tasks of k_iterate_single ran their solve in two stages (library built with -DAVBD_DF_STATS; per-wave counters of the last step)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
g = H.HipWorld(); g.scene(sys.argv[1] if len(sys.argv) > 1 else "Stress1000"); g.step(420); g.w.synchronize()
W = 1024
buf = np.zeros((W, 16), np.uint64)
g.pkg.lib().avbd_debug_iterate_stats(g.w.h, buf.ctypes.data_as(C.c_void_p), W)
act = buf[(buf[:, 5] > 0) & (buf[:, 5] < 64)].astype(np.float64)
print("waves %d, tasks %d, solved in two stages %d (%.1f %%)" % (len(act), act[:, 5].sum(), act[:, 15].sum(), 100.0 * act[:, 15].sum() / max(1.0, act[:, 5].sum())))
