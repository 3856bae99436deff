"""How many swept bodies of a scene have no swept neighbour (the fast sweep's isolated row), and the tiles they fill.  Usage: python tools/fast_isolated.py <scene> <steps>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
scene, steps = sys.argv[1], int(sys.argv[2])
g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1); g.step(steps)
b, f = g.bodies(), g.forces()
dyn = b["invMass"] > 0
a, bb = f["bodyA"], f["bodyB"]
both = dyn[a] & dyn[bb]
has = np.zeros(len(b), bool); has[a[both]] = True; has[bb[both]] = True
deg = np.zeros(len(b), int)
np.add.at(deg, a, f["numContacts"]); np.add.at(deg, bb, f["numContacts"])
iso = dyn & ~has
tiles_iso = sum(-(-int(((deg[iso] > lo) & (deg[iso] <= G)).sum()) // (64 // G)) for lo, G in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64))) + -(-int((deg[iso] == 0).sum()) // 64)
col = g.w.fast_colors()
print(scene, "step", steps, "plan", g.w.fast_plan_info(), "isolated bodies", int(iso.sum()), "of", int(dyn.sum()), "-> about", tiles_iso, "tiles; bodies per colour", np.bincount(col[dyn & has]))
