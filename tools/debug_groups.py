"""Per-wave time split of k_iterate_groups (library built with -DAVBD_DF_STATS, see tools/README.md)."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "StressN:20"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 410
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
t0 = time.perf_counter(); g.step(20); g.w.synchronize(); dt = (time.perf_counter() - t0) / 20
st = g.w.stats()
print("ms/step %.3f" % (dt * 1e3), "schedule", st["schedule"])
W = 4096
buf = np.zeros((W, 16), np.uint64)
g.pkg.lib().avbd_debug_iterate_stats(g.w.h, buf.ctypes.data_as(C.c_void_p), W)
act = buf[buf[:, 5] > 0].astype(np.float64)
print("active waves", len(act), "passes/wave %.1f" % act[:, 5].mean(), "tasks/wave %.1f" % act[:, 4].mean(), "batch %.2f" % (act[:, 4].sum() / act[:, 5].sum()))
print("wave total us: mean %.1f max %.1f" % ((act[:, 6] / 100).mean(), (act[:, 6] / 100).max()))
for i, n in ((0, "poll+idle"), (7, "finish: records, post, S"), (8, "finish: accumulate"), (9, "finish: solve, publish"), (1, "finish: dual"),
             (10, "prepare: constants, world inertia"), (11, "prepare: seeds, basis, Jacobians"), (12, "prepare: products of the first entry"), (13, "prepare: products of the second entry"), (2, "after prepare"), (3, "first prepare / multi: dual stores acknowledged"), (12, "multi: look at watched words (drained)"), (13, "multi: dual arithmetic + issue (not drained)"), (15, "multi: park issue (not drained)")):
    # (stamps inside a task are taken by lane 0 for ITS sub-group's tasks only: ~iterations of them per wave)
    print("  %-48s total us/wave %8.1f   per pass %6.2f   per task of lane 0's body %6.2f" % (n, (act[:, i] / 100).mean(), (act[:, i] / 100 / act[:, 5]).mean(), (act[:, i] / 100).mean() / 20.0))
print("poll rounds per wave %.0f" % act[:, 14].mean())
if act[:, 10].sum() > 0 and act[:, 10].sum() < 1e9:   # several bodies per sub-group, plain loop: o[10] tasks, o[11] tasks that were the oldest of their sub-group
    print("tasks %d, of which the oldest task of their sub-group when they ran: %.1f %%" % (act[:, 10].sum(), 100.0 * act[:, 11].sum() / act[:, 10].sum()))
