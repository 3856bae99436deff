"""ms/step of ONE scene sharded over n contexts of this process on one GPU (avbd_group_step: the schedule of the one-process-per-GPU
RCCL path with device copies instead of the collective; the contexts share the device's waves).  usage: time_sharded.py scene n presim"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "StressN:20"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
presim = int(sys.argv[3]) if len(sys.argv) > 3 else 410
axis = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pkg = H.load_product()
worlds = [H.HipWorld() for _ in range(n)]
for w in worlds:
    w.scene(scene)
b = worlds[0].bodies()
owner = pkg.partition_by_axis(b["pos"], b["invMass"], n, axis=axis)
for r, w in enumerate(worlds):
    w.w.set_ownership(n, r, owner)
ctx = [w.w for w in worlds]
pkg.group_step(ctx, presim)
for w in ctx: w.synchronize()
t0 = time.time(); steps = 50
pkg.group_step(ctx, steps)
for w in ctx: w.synchronize()
dt = (time.time() - t0) / steps * 1e3
st = [int(w.stats()["schedule"]) for w in ctx]
print(f"{scene} over {n} contexts (axis {axis}): {dt:.3f} ms/step, schedules {st}, owned {[int((owner == r).sum()) for r in range(n)]}")
