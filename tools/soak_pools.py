import sys, os
sys.path.insert(0, "/root/repo/tests")
import numpy as np, harness as H
import test_gpu_fuzz as F
variants = [dict(groups_waves=8), dict(groups_waves=4, group_pool_min_rounds=2), dict(groups_waves=12), dict(groups_waves=8, groups_rounds=16), dict(groups_waves=16, group_pool_min_rounds=2)]
bad = 0; ran = 0
for seed in range(1000, 1000 + int(sys.argv[1])):
    opt = variants[seed % len(variants)]
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        F.build_contacts_only(w, seed)
    g.w.set_option("groups", 1)
    for k, v in opt.items():
        g.w.set_option(k, v)
    try:
        for s in range(12):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1} with {opt}")
            ran += int(g.w.stats()["schedule"]) == 4
    except Exception as e:
        bad += 1; print("FAIL", seed, opt, str(e)[:300], flush=True)
    g.w.close()
print("seeds", sys.argv[1], "failures", bad, "group-schedule steps", ran)
