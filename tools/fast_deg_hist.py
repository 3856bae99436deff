"""Lanes per body the fast sweep's tiles would need: histogram of contacts per swept body (one lane per contact and endpoint) of a scene at a step.  Usage: python tools/fast_deg_hist.py <scene> <steps>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
scene, steps = sys.argv[1], int(sys.argv[2])
g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1); g.step(steps)
b, f = g.bodies(), g.forces()
dyn = b["invMass"] > 0
deg = np.zeros(len(b), int)
np.add.at(deg, f["bodyA"], f["numContacts"]); np.add.at(deg, f["bodyB"], f["numContacts"])
d = deg[dyn]
print(scene, "step", steps, "plan", g.w.fast_plan_info(), "swept", int(dyn.sum()), "lanes", int(d.sum()))
print("  contacts per body:", dict(zip(*np.unique(d, return_counts=True))))
for name, classes in (("8/16/32/64", (8, 16, 32, 64)), ("4/8/12/16/32/64", (4, 8, 12, 16, 32, 64)), ("8/12/16/32/64", (8, 12, 16, 32, 64)), ("8/12/16/20/32/64", (8, 12, 16, 20, 32, 64))):
    tiles = 0; lanes = 0
    rest = d[d > 0]
    lo = 0
    for G in classes:
        nb = int(((rest > lo) & (rest <= G)).sum()); per = 64 // G
        tiles += -(-nb // per); lo = G
    print(f"  classes {name}: about {tiles} tiles (+ colour fragmentation)")
# with the colours of the plan: tiles = sum over (colour, class) of ceil(bodies / bodies per tile), over a window of steps
for name, classes in (("8/16/32/64", (8, 16, 32, 64)), ("4/8/16/32/64", (4, 8, 16, 32, 64)), ("4/8/12/16/32/64", (4, 8, 12, 16, 32, 64))):
    counts = []
    for s in range(0, 60, 5):
        b, f = g.bodies(), g.forces()
        col = g.w.fast_colors()
        deg = np.zeros(len(b), int)
        np.add.at(deg, f["bodyA"], f["numContacts"]); np.add.at(deg, f["bodyB"], f["numContacts"])
        tiles = 0
        for c in range(col.max() + 1):
            dc = deg[(col == c)]
            tiles += -(-int((dc == 0).sum()) // 64)
            lo = 0
            for G in classes:
                nb = int(((dc > lo) & (dc <= G)).sum()); tiles += -(-nb // (64 // G)); lo = G
        counts.append((tiles, g.w.fast_plan_info()["tiles"], int(col.max()) + 1))
        g.step(5)
    print(f"  classes {name}: (tiles, the plan's own, colours) every 5 steps: {counts}")
