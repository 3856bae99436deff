"""Per-family device times (avbd_profile_*) of a scene's steady window in the exact and the fast mode.  Usage: python tools/fast_prof.py <scene> <warm> <steps> [key=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

scene, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[4:])
g = H.HipWorld(); g.scene(scene)
for k, v in opts.items():
    g.w.set_option(k, v)
g.step(warm); g.w.synchronize()
t0 = time.perf_counter(); g.step(steps); g.w.synchronize(); t1 = time.perf_counter()
print(f"{scene} {opts}: {(t1 - t0) / steps * 1e3:.4f} ms/step, schedule {int(g.w.stats()['schedule'])}, plan {g.w.fast_plan_info()}")
g.w.profile_enable(True); g.w.profile_reset()
g.step(steps); g.w.synchronize()
p = g.w.profile()
print("  per step, us: " + " ".join(f"{n[3:]}={float(p[n]) / steps * 1e3:.1f}" for n in p.dtype.names if n.startswith("ms_")))
st = g.w.stats()
print(f"  bodies {int(st['numDynamic'])} manifolds {int(st['numManifolds'])} contacts {int(st['numContacts'])} incidences {int(st['numIncidences'])}")
