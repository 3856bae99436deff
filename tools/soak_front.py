"""Random piles through the large-scene front end and topology kernels (k_narrowphase in both lane widths with the class order, k_scan / k_fill_adj /
k_sort_expand with register-ranked and insertion-sorted lists) against the oracle: python tools/soak_front.py <seeds> [first seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, harness as H
n_seeds = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0; most = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    o, g = H.CpuWorld("orc"), H.HipWorld()
    n = int(rng.integers(60, 160))
    for w in (o, g):
        r2 = np.random.default_rng(seed)
        if seed % 2:                                           # a platform under a heap of small boxes: more than sixteen forces (the insertion-sorted lists)
            w.add_body((40, 1, 40), 0.0, 0.6, (0, -0.5, 0))
            w.add_body((3.0, 0.4, 3.0), 1.0, 0.5, (0.0, 0.25, 0.0))
            H.random_pile(w, r2, 2 * n, spread=1.2, height=4.0, ground=False, size_range=(0.25, 0.5))
        else:
            H.random_pile(w, r2, n, spread=float(1.2 + 1.5 * (seed % 3)), height=5.0)
        w.set_params(iterations=5 + seed % 4)
    g.w.set_option("fused_front", 0); g.w.set_option("fused_topology", 0)
    g.w.set_option("narrow_lanes", 64 if seed % 4 < 2 else 32); g.w.set_option("narrow_order", 0 if seed % 7 == 0 else 1)
    try:
        for s in range(90 if seed % 2 else 60):
            o.step(); g.step()
            if s % 4 == 3:
                H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        f = g.forces()
        deg = np.bincount(np.concatenate([f["bodyA"][f["bodyA"] >= 0], f["bodyB"]]), minlength=2 * n + 3)
        most = max(most, int(deg[1:].max()))
    except Exception as e:
        bad += 1; print("FAIL", seed, str(e)[:400], flush=True)
    g.w.close()
print("seeds", n_seeds, "failures", bad, "longest list of a swept body", most)
