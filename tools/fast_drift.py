import os, sys
import numpy as np
sys.path.insert(0, "tests")
import harness as H
for scene in ("Pyramid", "Wall"):
    ws = {}
    for name, opts in (("exact", {}), ("fast", {"sweep": 1}), ("fast_recolor", {"sweep": 1, "fast_recolor": 1})):
        g = H.HipWorld(); g.scene(scene)
        for k, v in opts.items(): g.w.set_option(k, v)
        b0 = g.bodies()["pos"].copy()
        row = []
        for s in range(6):
            g.step(100)
            b = g.bodies()
            dyn = b["invMass"] > 0
            d = b["pos"][dyn] - b0[dyn]
            v = np.linalg.norm(b["linVel"][dyn], axis=1).max()
            row.append((float(np.abs(d[:, [0, 2]]).max()), float(np.abs(d[:, 1]).max()), float(v)))
        ws[name] = row
        print(scene, name, " ".join(f"[xz {a:.4f} y {b:.4f} v {c:.4f}]" for a, b, c in row))
