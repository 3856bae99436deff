"""A/B of environment switches on one scene (product build): every variant runs in its own process, several repeats, interleaved.

    python tools/ab_env.py <scene> <presim> <steps> <repeats> VAR=a,b[,c] [VAR2=x,y]      (cartesian product of the values)

Prints ms/step (wall, steps queued back to back) and the iteration launch's mean duration from the library's HIP events."""
import itertools, json, os, subprocess, sys

CHILD = r'''
import sys, time, os, json, ctypes as C
sys.path.insert(0, %(root)r)
import importlib
import numpy as np
pkg = importlib.import_module("avbd-demo3d_amd")
scene, presim, steps = %(scene)r, %(presim)d, %(steps)d
w = pkg.World(0); w.scene(scene); w.step(presim); w.synchronize()
w.step(20); w.synchronize()
t0 = time.perf_counter(); w.step(steps); w.synchronize(); dt = (time.perf_counter() - t0) / steps
w.profile_enable(True); w.profile_reset(); w.step(20); w.synchronize(); p = w.profile(); w.profile_enable(False)
st = w.stats()
info = (C.c_int * 4)()
L = pkg.lib()
if hasattr(L, "avbd_debug_place_info"): L.avbd_debug_place_info(w.h, info, None)
print(json.dumps({"ms": dt * 1e3, "iter_ms": float(p["ms_primal"]) / 20, "front_ms": float(p["ms_narrowphase"] + p["ms_broadphase"]) / 20, "topo_ms": float(p["ms_topology"]) / 20,
                  "tail_ms": float(p["ms_velocity"]) / 20, "schedule": int(st["schedule"]), "place": list(info)}))
w.close()
'''

def main():
    scene, presim, steps, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[5:]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    combos = list(itertools.product(*[v for _, v in axes])) or [()]
    res = {c: [] for c in combos}
    for r in range(reps):
        for c in combos:
            env = dict(os.environ)
            for (k, _), v in zip(axes, c):
                env[k] = v
            out = subprocess.run([sys.executable, "-c", CHILD % {"root": root, "scene": scene, "presim": presim, "steps": steps}], env=env, capture_output=True, text=True)
            try:
                res[c].append(json.loads(out.stdout.strip().splitlines()[-1]))
            except Exception:
                print("FAILED", c, out.stdout[-500:], out.stderr[-1500:], flush=True)
    for c in combos:
        tag = " ".join(f"{k}={v}" for (k, _), v in zip(axes, c))
        ms = sorted(x["ms"] for x in res[c]); it = sorted(x["iter_ms"] for x in res[c])
        if not ms: continue
        last = res[c][-1]
        print(f"{scene} {tag:40s} ms/step min {ms[0]:.4f} med {ms[len(ms)//2]:.4f} | iterate min {it[0]:.4f} med {it[len(it)//2]:.4f} | front {last['front_ms']:.4f} topo {last['topo_ms']:.4f} tail {last['tail_ms']:.4f} "
              f"schedule {last['schedule']} place {last['place']}", flush=True)

if __name__ == "__main__":
    main()
