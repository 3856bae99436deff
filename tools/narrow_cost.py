"""k_narrowphase per manifold in a homogeneous scene (an aligned grid: every manifold a four-point face manifold) against a settled random pile (edge
contacts and one- to four-point manifolds mixed in every wave): what does the mix cost?   python tools/narrow_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
pkg = importlib.import_module("avbd-demo3d_amd")
for scene, presim in (("PackedGrid:40,40,40", 20), ("StressN:40", 470), ("PackedGrid:20,20,20", 20), ("StressN:20", 410)):
    w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
    w.profile_enable(True); w.profile_reset(); w.step(10); w.synchronize(); p = w.profile(); w.profile_enable(False)
    st = w.stats()
    f = w.forces()
    hist = np.bincount(f["numContacts"], minlength=5)[1:5]
    edge = int(((f["contacts"]["feature"][:, 0] >> 24) == 2).sum())
    print(f"{scene}: manifolds {int(st['numManifolds'])} candidates {int(st['numCandidates'])} contacts {int(st['numContacts'])}  narrow {float(p['ms_narrowphase']) / 10 * 1e3:.1f} us "
          f"= {float(p['ms_narrowphase']) / 10 * 1e6 / max(1, int(st['numManifolds'])):.2f} ns per manifold; contacts per manifold 1/2/3/4: {hist.tolist()}, edge contacts {edge}", flush=True)
    w.close()
