"""Random scenes with joints, springs, ignore markers, tilted gravity, random solver parameters (tests/test_gpu_fuzz.py::build) and forces / bodies added while
they run, through whatever schedule the library picks, against the oracle every second step: python tools/soak_mixed.py <seeds> [first seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, harness as H
import test_gpu_fuzz as F
n_seeds = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
bad = 0; sched = {}
for seed in range(first, first + n_seeds):
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        n = F.build(w, seed)
    r = np.random.default_rng(seed + 7)
    try:
        for s in range(36):
            o.step(); g.step()
            if s % 2 == 1: H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
            if s in (9, 21) and n > 4:
                a, b = int(r.integers(1, n)), int(r.integers(1, n))
                kind = int(r.integers(0, 4))
                for w in (o, g):
                    if kind == 0 and a != b: w.add_joint(a, b, (0.3, 0, 0), (-0.3, 0, 0))
                    elif kind == 1 and a != b: w.add_spring(a, b, (0, 0.2, 0), (0, -0.2, 0), 800.0)
                    elif kind == 2 and a != b: w.add_ignore(a, b)
                    else: w.add_body((0.7, 0.5, 0.9), 1.2, 0.4, (0.3, 8.0, -0.2))
        k = int(g.w.stats()["schedule"]); sched[k] = sched.get(k, 0) + 1
    except Exception as e:
        if "pool overflow" in str(e): sched["too dense"] = sched.get("too dense", 0) + 1
        else: bad += 1; print("FAIL", seed, str(e)[:400], flush=True)
    g.w.close()
print("seeds", n_seeds, "failures", bad, "schedules at the end", sched)
