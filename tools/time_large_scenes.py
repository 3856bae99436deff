import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("avbd-demo3d_amd")
for scene, presim in (("StressN:20", 400), ("StressN:40", 450)):
    w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
    t0 = time.perf_counter(); w.step(20); w.synchronize(); dt = (time.perf_counter() - t0) / 20
    w.profile_enable(True); w.profile_reset(); w.step(10); w.synchronize(); p = w.profile(); w.profile_enable(False)
    st = w.stats()
    print(scene, "ms/step %.3f" % (dt * 1e3), "schedule", st["schedule"], "contacts", st["numContacts"], "manifolds", st["numManifolds"],
          {k: round(float(p[k]) / 10, 3) for k in p.dtype.names if k.startswith("ms_")} if hasattr(p, "dtype") else {k: round(float(v) / 10, 3) for k, v in p.items() if k.startswith("ms_")}, flush=True)
    w.close()
