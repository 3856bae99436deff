"""Random piles through exact cuts between the contexts of one process (avbd_group_step, csrc/avbd_cuts.h) against the single-GPU sweep of the same
library — which tools/soak_groups.py holds against the oracle: python tools/soak_exact.py <seeds> [first seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, harness as H
n_seeds = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
pkg = H.load_product()
bad = 0; declined = 0; ran = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    nranks = (2, 3, 4)[seed % 3]; axis = (0, 2)[(seed // 3) % 2]
    n = int(rng.integers(120, 500))
    def build(w):
        r2 = np.random.default_rng(seed)
        if seed % 4 == 1:
            w.add_body((40, 1, 40), 0.0, 0.6, (0, -0.5, 0))
            w.add_body((3.0, 0.4, 3.0), 1.0, 0.5, (0.0, 0.25, 0.0))
            H.random_pile(w, r2, n, spread=1.5, height=4.0, ground=False, size_range=(0.25, 0.5))
        else:
            H.random_pile(w, r2, n, spread=float(2.0 + 1.5 * (seed % 3)), height=5.0)
        w.set_params(iterations=5 + seed % 4)
    os.environ["AVBD_SHARD_EXACT"] = "1"
    if seed % 5 == 0: os.environ["AVBD_GROUPS_WAVES"] = "8"
    worlds = [H.HipWorld() for _ in range(nranks)]
    for w in worlds: build(w)
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], nranks, axis=axis)
    for r, w in enumerate(worlds): w.w.set_ownership(nranks, r, owner)
    os.environ.pop("AVBD_SHARD_EXACT"); os.environ.pop("AVBD_GROUPS_WAVES", None)
    single = H.HipWorld(); build(single)
    try:
        for s in range(10):
            pkg.group_step([w.w for w in worlds], 5); single.step(5)
            bs = single.bodies()
            for r, w in enumerate(worlds):
                assert H.same_bits(bs, w.bodies()), f"rank {r} of {nranks} (axis {axis}): body state differs from the single-GPU sweep after {5 * (s + 1)} steps"
        ran += 1
    except Exception as e:
        msg = str(e)
        if "pool overflow" in msg or "cannot run them" in msg: declined += 1     # (a body with more than 64 contacts: a refusal, by the message)
        else: bad += 1; print("FAIL", seed, msg[:400], flush=True)
    for w in worlds + [single]: w.w.close()
print("seeds", n_seeds, "failures", bad, "ran to the end", ran, "declined / too dense", declined)
