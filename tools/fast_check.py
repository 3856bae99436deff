"""Developer check of the fast sweep (option "sweep" = 1) on the GPU box: the same scene in the exact mode and in the fast mode
(stamped records / grid barrier per colour), teacher-forced single steps, the colouring's statistics, ms per step.
Usage: python tools/fast_check.py [scene ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H


def rot_angle(qa, qb):
    d = np.abs(np.sum(qa.astype(np.float64) * qb.astype(np.float64), axis=1))
    return 2.0 * np.arccos(np.clip(d, 0.0, 1.0))


def diff(a, b):
    ba, bb = a.bodies(), b.bodies()
    dp = np.linalg.norm(ba["pos"].astype(np.float64) - bb["pos"].astype(np.float64), axis=1)
    return float(dp.max()), float(rot_angle(ba["quat"], bb["quat"]).max())


def world(scene, **opts):
    g = H.HipWorld(); g.scene(scene)
    for k, v in opts.items():
        g.w.set_option(k, v)
    return g


def teacher_forced(scene, at, steps=(1, 10)):
    ex = world(scene); ex.step(at)
    b, f = ex.bodies(), ex.forces()
    out = []
    for stamps in (1, 0):
        fa = world(scene, sweep=1, fast_stamps=stamps)
        fa.set_bodies(b); fa.set_forces(f); fa.set_step_index(at)
        e2 = world(scene); e2.set_bodies(b); e2.set_forces(f); e2.set_step_index(at)
        done = 0
        row = []
        for s in steps:
            fa.step(s - done); e2.step(s - done); done = s
            row.append(diff(e2, fa))
        info = fa.w.fast_plan_info()
        out.append((stamps, row, info, int(fa.w.stats()["schedule"]), fa.bodies()))
        fa.close(); e2.close()
    same = H.same_bits(out[0][4], out[1][4])
    ex.close()
    for stamps, row, info, sched, _ in out:
        print(f"  {scene} @ {at} stamps={stamps} schedule={sched} plan={info}: " + ", ".join(f"dpos {p:.3e} m drot {r:.3e} rad" for p, r in row))
    print(f"  stamps vs barrier bit-identical: {same}")


def timing(scene, warm, steps, **opts):
    g = world(scene, **opts)
    g.step(warm); g.w.synchronize()
    t0 = time.perf_counter(); g.step(steps); g.w.synchronize(); t1 = time.perf_counter()
    st = g.w.stats()
    info = g.w.fast_plan_info() if opts.get("sweep") else {}
    d = g.diag()
    print(f"  {scene} {opts}: {(t1 - t0) / steps * 1e3:.4f} ms/step schedule={int(st['schedule'])} manifolds={int(st['numManifolds'])} contacts={int(st['numContacts'])} maxPen={float(d['maxPenetration']):.4f} {info}")
    g.close()


if __name__ == "__main__":
    scenes = sys.argv[1:] or ["Stack", "Pyramid"]
    for sc in scenes:
        print(sc)
        teacher_forced(sc, 60)
    print("Stress1000")
    teacher_forced("Stress1000", 200, steps=(1, 10))
    teacher_forced("Stress1000", 400, steps=(1, 10))
    timing("Stress1000", 400, 200)
    timing("Stress1000", 400, 200, sweep=1)
    timing("Stress1000", 400, 200, sweep=1, fast_stamps=0)
