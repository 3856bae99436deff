"""What the 10 000 welds of BASELINE configs[4] cost on one GPU: the 64 000-cube pile with and without them, and without them on the schedule the welds force
(the resident schedule with parked bodies, `groups` 0).   python tools/joint_cost.py [presim=470]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("avbd-demo3d_amd")
presim = int(sys.argv[1]) if len(sys.argv) > 1 else 470
for joints, opts in ((10000, {}), (0, {"groups": 0}), (0, {})):
    w = pkg.World(0); w.scene("StressN:40")
    for k in range(joints):
        w.add_joint(1 + 2 * k, 2 + 2 * k, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
    for k, v in opts.items():
        w.set_option(k, v)
    w.step(presim); w.synchronize()
    t0 = time.perf_counter(); w.step(20); w.synchronize(); dt = (time.perf_counter() - t0) / 20
    w.profile_enable(True); w.profile_reset(); w.step(10); w.synchronize(); p = w.profile(); w.profile_enable(False)
    st = w.stats()
    print(f"joints {joints} {opts}: {dt * 1e3:.3f} ms/step, schedule {int(st['schedule'])}, manifolds {int(st['numManifolds'])}",
          {k: round(float(p[k]) / 10, 3) for k in p.dtype.names if k.startswith("ms_")}, flush=True)
    w.close()
