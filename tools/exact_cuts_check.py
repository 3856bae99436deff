"""Steps a scene on N contexts of one GPU with exact cuts (AVBD_SHARD_EXACT=1) beside a single context and reports, every five steps,
whether all ranks hold the single-GPU bits.    python tools/exact_cuts_check.py <scene> <ranks> <steps> [axis]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
axis = int(sys.argv[4]) if len(sys.argv) > 4 else 0
os.environ["AVBD_SHARD_EXACT"] = "1"
pkg = H.load_product()
worlds = [H.HipWorld() for _ in range(n)]
for w in worlds: w.scene(scene)
b = worlds[0].bodies()
owner = pkg.partition_by_axis(b["pos"], b["invMass"], n, axis=axis)
for r, w in enumerate(worlds): w.w.set_ownership(n, r, owner)
single = H.HipWorld(); single.scene(scene)
done = 0; tg = ts = 0.0
try:
    for s in range(0, steps, 5):
        t0 = time.perf_counter(); pkg.group_step([w.w for w in worlds], 5)
        for w in worlds: w.w.synchronize()
        t1 = time.perf_counter(); single.step(5); single.w.synchronize(); t2 = time.perf_counter()
        tg, ts = t1 - t0, t2 - t1
        done = s + 5
        same = all(H.same_bits(single.bodies(), w.bodies()) for w in worlds)
        if not same or s % 25 == 20: print(scene, n, "ranks, step", done, "bit-identical:", same, "manifolds", int(single.w.stats()["numManifolds"]), "schedules", [int(w.w.stats()["schedule"]) for w in worlds],
                                           "ms/step: %d ranks %.2f, one context %.2f" % (n, tg / 5 * 1e3, ts / 5 * 1e3), flush=True)
        if not same: break
except Exception as e:
    print(scene, n, "FAILED after step", done, str(e)[:400])
