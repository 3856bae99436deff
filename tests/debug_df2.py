import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import harness as H
for iters in (1, 2, 3):
    o = H.CpuWorld("orc"); o.scene("Stack"); o.set_params(iterations=iters)
    g = H.HipWorld(); g.scene("Stack"); g.set_params(iterations=iters)
    o.step(); g.step()
    bo, bg = o.bodies(), g.bodies()
    fo, fg = o.forces(), g.forces()
    print("iters", iters, "pos1 orc", bo["pos"][1], "hip", bg["pos"][1], "same", H.same_bits(bo["pos"], bg["pos"]))
    print("   lam orc", fo["lambda"][0][:3], "hip", fg["lambda"][0][:3], " pen orc", fo["penalty"][0][:3], "hip", fg["penalty"][0][:3])
