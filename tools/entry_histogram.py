import sys, os
sys.path.insert(0, "/root/repo/tests")
import numpy as np, harness as H
scene = sys.argv[1]; pre = int(sys.argv[2])
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
f = g.forces(); b = g.bodies()
dyn = b["invMass"] > 0
nc = f["numContacts"].astype(np.int64)
ent = np.zeros(len(b), np.int64)
np.add.at(ent, f["bodyA"], nc); np.add.at(ent, f["bodyB"], nc)
e = ent[dyn]
print(scene, "bodies", dyn.sum(), "entries mean %.2f" % e.mean(), "<=8: %d  9-16: %d  17-32: %d  >32: %d  max %d" % ((e <= 8).sum(), ((e > 8) & (e <= 16)).sum(), ((e > 16) & (e <= 32)).sum(), (e > 32).sum(), e.max()))
