"""Random piles through the small scenes' fused kernels with their defaults (k_front_small: the cooperative narrow phase; k_topology_small: one pass over a
body's keys, the kept sweep order and wave placement, predictor on the other workgroups; k_iterate_single or the group schedule behind them) against the
oracle, with bodies added while the placement is kept: python tools/soak_small.py <seeds> [first seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, harness as H
n_seeds = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
bad = 0; sched = {}
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    if os.environ.get("SOAK_VERBOSE"): print("seed", seed, flush=True)
    o, g = H.CpuWorld("orc"), H.HipWorld()
    n = int(rng.integers(40, 140)) * int(os.environ.get("SOAK_SCALE", "1"))   # SOAK_SCALE=4: heaps with more entries than the LDS staging of k_topology_small holds
    for w in (o, g):
        r2 = np.random.default_rng(seed)
        if seed % 3 == 1:                                      # a platform under a heap of small boxes: long lists, more entries than the LDS staging holds
            w.add_body((40, 1, 40), 0.0, 0.6, (0, -0.5, 0))
            w.add_body((3.0, 0.4, 3.0), 1.0, 0.5, (0.0, 0.25, 0.0))
            H.random_pile(w, r2, 2 * n, spread=1.2, height=4.0, ground=False, size_range=(0.25, 0.5))
        else:
            H.random_pile(w, r2, n, spread=float(1.2 + 1.5 * (seed % 3)), height=5.0)
        w.set_params(iterations=5 + seed % 4)
    g.w.set_option("place_every", (1, 3, 8, 50)[seed % 4])
    try:
        for s in range(80):
            o.step(); g.step()
            if s in (17, 41) and seed % 2 == 0:                # one more swept body (or a static one) while a placement is kept
                for w in (o, g):
                    w.add_body((0.8, 0.8, 0.8), 0.0 if s == 41 and seed % 4 == 0 else 1.5, 0.5, (0.1 * (seed % 7), 7.0 + 0.01 * s, -0.2))
            if s % 4 == 3:
                H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        k = int(g.w.stats()["schedule"]); sched[k] = sched.get(k, 0) + 1
    except Exception as e:
        if "pool overflow" in str(e): sched["pile too dense for the pools"] = sched.get("pile too dense for the pools", 0) + 1   # AVBD_E_CAPACITY: an error, never a fault
        else: bad += 1; print("FAIL", seed, str(e)[:400], flush=True)
    g.w.close()
print("seeds", n_seeds, "failures", bad, "schedules", sched)
