import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import harness as H
scene = sys.argv[1]; steps = int(sys.argv[2]); check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = H.HipWorld(); g.scene(scene)
t0=time.perf_counter(); g.step(steps); g.w.synchronize(); dt=time.perf_counter()-t0
st=g.w.stats()
print(scene, "gpu ms/step", dt/steps*1e3, "stats", st)
t0=time.perf_counter(); g.step(steps); g.w.synchronize(); dt=time.perf_counter()-t0
print(scene, "gpu ms/step (second window)", dt/steps*1e3, "stats", g.w.stats())
if check:
    o = H.CpuWorld("orc"); o.scene(scene)
    t0=time.perf_counter(); o.step(2*steps); dt=time.perf_counter()-t0
    print("oracle ms/step", dt/(2*steps)*1e3)
    print("bit-exact bodies:", H.same_bits(o.bodies(), g.bodies()))
