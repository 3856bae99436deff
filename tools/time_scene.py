import sys, time, os
sys.path.insert(0, "/root/repo")
import importlib
pkg = importlib.import_module("avbd-demo3d_amd")
for scene, presim in ((sys.argv[1], int(sys.argv[2])),):
    w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
    t0 = time.perf_counter(); w.step(50); w.synchronize(); dt = (time.perf_counter() - t0) / 50
    w.profile_enable(True); w.profile_reset(); w.step(10); w.synchronize(); p = w.profile(); w.profile_enable(False)
    st = w.stats()
    print(scene, "ms/step %.3f" % (dt * 1e3), "schedule", st["schedule"], "contacts", st["numContacts"], "manifolds", st["numManifolds"],
          {k: round(float(p[k]) / 10, 3) for k in p.dtype.names if k.startswith("ms_")} if hasattr(p, "dtype") else {k: round(float(v) / 10, 3) for k, v in p.items() if k.startswith("ms_")}, flush=True)
    w.close()
