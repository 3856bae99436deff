"""The fast sweep's all-gather form on ONE GPU: the 64 000-cube scene as 8 x-slabs (avbd_group_step: device copies instead of RCCL), the ranks' launches one after
the other on the device — per-rank device time of the iteration launches at 8 000 owned bodies, next to the one-GPU fast line.  UNMEASURED ON MULTI-GPU HARDWARE:
what this gives is the per-rank kernel time an 8-GPU run would overlap.  Usage: python tools/time_sharded_fast.py [scene] [ranks] [presim] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

scene = sys.argv[1] if len(sys.argv) > 1 else "StressN:40"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
presim = int(sys.argv[3]) if len(sys.argv) > 3 else 470
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
pkg = H.load_product()
# presimulate once on one context in the fast mode, then hand the state to the ranks
one = H.HipWorld(); one.scene(scene); one.w.set_option("sweep", 1); one.step(presim)
b, f, si = one.bodies(), one.forces(), one.step_index()
one.step(5); one.w.synchronize()
t0 = time.perf_counter(); one.step(steps); one.w.synchronize(); t1 = time.perf_counter()
print(f"{scene}: one GPU, fast sweep: {(t1 - t0) / steps * 1e3:.3f} ms/step", one.w.fast_plan_info())
owner = pkg.partition_by_axis(b["pos"], b["invMass"], n, axis=0)
worlds = []
for r in range(n):
    w = H.HipWorld(); w.scene(scene); w.w.set_option("sweep", 1)
    w.w.set_ownership(n, r, owner)                     # (before the first step; the slabs are those of the landed pile)
    mine = (owner[np.maximum(f["bodyA"], 0)] == r) | (owner[np.maximum(f["bodyB"], 0)] == r)      # the forces a rank of a sharded run holds: those that touch one of its bodies
    w.set_bodies(b); w.set_forces(f[mine]); w.set_step_index(si)
    worlds.append(w)
pkg.group_step([w.w for w in worlds], 3)
for w in worlds:
    w.w.profile_enable(True); w.w.profile_reset()
t0 = time.perf_counter(); pkg.group_step([w.w for w in worlds], steps)
for w in worlds:
    w.w.synchronize()
t1 = time.perf_counter()
print(f"{n} ranks on one device, serialised: {(t1 - t0) / steps * 1e3:.3f} ms/step wall (all ranks one after the other + host-side exchange)")
for r, w in enumerate(worlds):
    p = w.w.profile(); st = w.w.stats()
    per = {k[3:]: float(p[k]) / steps for k in p.dtype.names if k.startswith("ms_")}
    print(f"  rank {r}: owned {int((owner == r).sum())} bodies, manifolds {int(st['numManifolds'])}, schedule {int(st['schedule'])}, plan {w.w.fast_plan_info()}; per step ms: " + " ".join(f"{k}={v:.3f}" for k, v in per.items()) + f"  sum={sum(per.values()):.3f}")
