"""Colours of the fast sweep's plan over a window of steps, and the time per step, for option settings given as key=value.  Usage: python tools/fast_colour_hist.py <scene> <warm> <steps> [key=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
scene, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[4:])
g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1)
for k, v in opts.items():
    g.w.set_option(k, v)
g.step(warm); g.w.synchronize()
t0 = time.perf_counter(); g.step(steps); g.w.synchronize(); t1 = time.perf_counter()
cols = []
for s in range(steps):
    g.step(); cols.append(g.w.fast_plan_info()["colors"])
g.w.profile_enable(True); g.w.profile_reset(); g.step(steps); g.w.synchronize(); p = g.w.profile()
print(f"{scene} {opts}: {(t1 - t0) / steps * 1e3:.4f} ms/step over steps {warm}..{warm + steps}; colours in the next {steps} steps: {dict(zip(*np.unique(cols, return_counts=True)))}; primal {float(p['ms_primal']) / steps * 1e3:.1f} topology {float(p['ms_topology']) / steps * 1e3:.1f} us")
