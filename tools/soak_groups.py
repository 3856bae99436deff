"""Random piles through the group schedule (k_group_plan + k_iterate_groups in its one-body and several-bodies forms, pools, arrival counters, the
three lane classes, plans that decline) against the oracle: python tools/soak_groups.py <seeds> [first seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, harness as H
n_seeds = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0; sched = {}
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    o, g = H.CpuWorld("orc"), H.HipWorld()
    n = int(rng.integers(60, 260)) * int(os.environ.get("SOAK_SCALE", "1"))
    for w in (o, g):
        r2 = np.random.default_rng(seed)
        if seed % 3 == 1:                                      # a platform under a heap of small boxes: bodies with 17..64 entries (the 32- and 64-lane classes) and beyond (declined)
            w.add_body((40, 1, 40), 0.0, 0.6, (0, -0.5, 0))
            w.add_body((3.0, 0.4, 3.0), 1.0, 0.5, (0.0, 0.25, 0.0))
            H.random_pile(w, r2, n, spread=1.2, height=4.0, ground=False, size_range=(0.25, 0.5))
        else:
            H.random_pile(w, r2, n, spread=float(1.5 + 1.5 * (seed % 3)), height=5.0)
        w.set_params(iterations=5 + seed % 4)
    g.w.set_option("groups", 1); g.w.set_option("groups_waves", (8, 16, 32, 64, 0)[seed % 5])
    if seed % 7 == 0: g.w.set_option("group_pool", 0)
    if seed % 11 == 0: g.w.set_option("group_dep", 0)
    try:
        for s in range(72):
            o.step(); g.step()
            if s % 4 == 3:
                H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        k = int(g.w.stats()["schedule"]); sched[k] = sched.get(k, 0) + 1
    except Exception as e:
        if "pool overflow" in str(e): sched["pile too dense for the pools"] = sched.get("pile too dense for the pools", 0) + 1
        else: bad += 1; print("FAIL", seed, str(e)[:400], flush=True)
    g.w.close()
print("seeds", n_seeds, "failures", bad, "schedules at the end", sched)
