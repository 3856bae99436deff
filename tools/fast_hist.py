import sys
sys.path.insert(0,'tests')
import harness as H, numpy as np
for scene, at in (("Stress1000",500),("StressN:20",430),("StressN:40",480)):
    g=H.HipWorld(); g.scene(scene); g.w.set_option("sweep",1); g.step(at)
    b=g.bodies(); f=g.forces(); dyn=b["invMass"]>0
    cnt=np.zeros(len(b),int)
    np.add.at(cnt, f["bodyA"], f["numContacts"]); np.add.at(cnt, f["bodyB"], f["numContacts"])
    c=cnt[dyn]
    h=np.bincount(np.minimum(c,40))
    print(scene, "mean", c.mean(), "hist 0..", h.tolist())
    for lo,hi in ((0,0),(1,4),(5,8),(9,12),(13,16),(17,32),(33,64)):
        print("   ",lo,hi, int(((c>=lo)&(c<=hi)).sum()))
    g.close()
