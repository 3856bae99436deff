import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
pkg = importlib.import_module("avbd-demo3d_amd")
os.makedirs("gpurun_out/graphs", exist_ok=True)
for scene, presim, tag in (("Stress1000", 420, "s10"), ("StressN:20", 410, "s20"), ("StressN:40", 460, "s40")):
    w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
    f = w.forces(); b = w.bodies()
    np.savez_compressed(f"gpurun_out/graphs/{tag}.npz", a=f["bodyA"].astype(np.int32), b=f["bodyB"].astype(np.int32), nc=f["numContacts"].astype(np.int8), inv=b["invMass"])
    print(tag, len(f), flush=True)
    w.close()
