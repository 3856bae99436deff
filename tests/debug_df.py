import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "Stack"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
o = H.CpuWorld("orc"); o.scene(scene)
g = H.HipWorld(); g.scene(scene)
for s in range(steps):
    o.step(); g.step()
    bo, bg = o.bodies(), g.bodies()
    bad = np.nonzero(np.any(bo["pos"].view(np.uint32) != bg["pos"].view(np.uint32), axis=1))[0]
    print(f"step {s+1}: mismatching bodies {bad[:10]} of {len(bo)}; stats", g.w.stats())
    if len(bad):
        i = bad[0]
        print("  orc", bo["pos"][i], bo["quat"][i]); print("  hip", bg["pos"][i], bg["quat"][i])
        fo, fg = o.forces(), g.forces()
        print("  forces", len(fo), len(fg))
        if len(fo):
            print("  orc lambda", fo["lambda"][0][:12], "\n  hip lambda", fg["lambda"][0][:12])
            print("  orc pen", fo["penalty"][0][:12], "\n  hip pen", fg["penalty"][0][:12])
        break
