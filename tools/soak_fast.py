"""Soak of the fast sweep (round 6: one-XCD form, isolated row, 4-lane class, split second pass): long runs of the small scenes and many random scenes, every step checked for a
time-out / veto / non-finite state; a second world with the barrier form (or the device-wide protocol) must hold the same bits.  Usage: python tools/soak_fast.py [steps] [seeds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
from test_gpu_fuzz import build as build_random
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
for scene, other in (("Stress1000", {"fast_one_xcd": 0}), ("StressN:9", {"fast_stamps": 0}), ("Pyramid", {"debug_fast_xcd_mismatch": 1}), ("Stack", {"fast_form": 3})):
    a, b = H.HipWorld(), H.HipWorld()
    for w, o in ((a, {}), (b, other)):
        w.scene(scene); w.w.set_option("sweep", 1)
        for k, v in o.items(): w.w.set_option(k, v)
    chunk = 250
    for s in range(0, steps, chunk):
        a.step(chunk); b.step(chunk)
        ba, bb = a.bodies(), b.bodies()
        assert H.same_bits(ba, bb), (scene, s)
        assert np.isfinite(ba["pos"]).all() and np.isfinite(ba["quat"]).all(), (scene, s)
        info = a.w.fast_plan_info()
        assert info["bad"] == 0 and int(a.w.stats()["schedule"]) == 5, (scene, s, info)
    print(f"{scene}: {steps} steps ok, same bits as {other}; plan {a.w.fast_plan_info()}  [{time.time() - t0:.0f} s]"); sys.stdout.flush()
    a.close(); b.close()
bad = 0
for seed in range(9000, 9000 + seeds):
    a, b = H.HipWorld(), H.HipWorld()
    for w, o in ((a, {}), (b, {"fast_one_xcd": 0, "fast_form": 2 if seed % 2 else 3})):
        build_random(w, seed); w.w.set_option("sweep", 1)
        for k, v in o.items(): w.w.set_option(k, v)
    for s in range(40):
        a.step(); b.step()
    ba, bb = a.bodies(), b.bodies()
    assert H.same_bits(ba, bb), seed
    assert np.isfinite(ba["pos"]).all(), seed
    sched = int(a.w.stats()["schedule"])
    if sched != 5: bad += 1
    a.close(); b.close()
print(f"{seeds} random scenes x 40 steps ok ({bad} ended on an exact schedule: vetoed or ineligible)  [{time.time() - t0:.0f} s]")
