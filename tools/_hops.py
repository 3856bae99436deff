import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
pkg = importlib.import_module("avbd-demo3d_amd")
import bench
for scene, presim in (("Stress1000", 420), ("StressN:20", 400), ("StressN:40", 450)):
    w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
    t0 = time.perf_counter(); w.step(10); w.synchronize(); dt = (time.perf_counter() - t0) / 10
    f = w.forces(); b = w.bodies()
    hops = bench.dependency_chain_hops(f, b, 20)
    deg = np.zeros(len(b), int)
    for a, c in zip(f["bodyA"], f["bodyB"]):
        if a >= 0 and c >= 0: deg[a] += 1; deg[c] += 1
    print(scene, "ms/step %.3f" % (dt * 1e3), "hops", hops, "us/hop %.2f" % (dt * 1e6 / hops), "mean manifolds/body %.2f max %d" % (deg[1:].mean(), deg[1:].max()), flush=True)
    w.close()
