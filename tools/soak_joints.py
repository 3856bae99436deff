"""Random jointed scenes (hard, soft and world-anchored joints, ignore markers: tests/test_gpu_fuzz.py::build_jointed) through the JOINT-aware kernels of the group schedule
(k_iterate_groups<*, 0, true>, round 5) against the oracle, every step bit for bit: one body per sub-group, pools, a task in hand, the plain several-bodies loop, with and
without arrival counters.      python tools/soak_joints.py <seeds>"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, harness as H
import test_gpu_fuzz as F
variants = [dict(), dict(groups_waves=3), dict(groups_waves=3, group_pool=0, group_spec=2), dict(groups_waves=4, group_pool=0, group_spec=1, group_spec_rounds=16),
            dict(groups_waves=2, group_pool_min_rounds=2), dict(groups_waves=3, group_pool=0, group_dep=0), dict(groups_waves=6, groups_rounds=16)]
bad = 0; ran = 0
for seed in range(5000, 5000 + int(sys.argv[1])):
    opt = variants[seed % len(variants)]
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        F.build_jointed(w, seed)
    g.w.set_option("groups", 1)
    for k, v in opt.items():
        g.w.set_option(k, v)
    try:
        for s in range(14):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1} with {opt}")
            ran += int(g.w.stats()["schedule"]) == 4
    except Exception as e:
        bad += 1; print("FAIL", seed, opt, str(e)[:300], flush=True)
    g.w.close()
print("seeds", sys.argv[1], "failures", bad, "group-schedule steps", ran)
