"""Measures what DESIGN.md section 2b states about the fast sweep (option "sweep" = 1) — the numbers tests/test_gpu_fast.py bounds:
 (i)   one teacher-forced step from every committed reference snapshot and from the exact mode's own state of the bench scenes: max |dpos| (m), max rotation
       angle (rad) against the reference / the exact mode, and their growth over 10 steps;
 (ii)  invariants of a 600-step Stress1000 run against the reference's diagnostics (tests/golden/stress1000_diag_600.npy), Stack after 300 steps;
 (iii) bit-identical results from run to run.
Usage: python tools/fast_tolerance.py [--big]   (--big adds StressN:20 / StressN:40)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def rot_angle(qa, qb):
    d = np.abs(np.sum(qa.astype(np.float64) * qb.astype(np.float64), axis=1))
    return 2.0 * np.arccos(np.clip(d, 0.0, 1.0))


def body_diff(ba, bb):
    dp = np.linalg.norm(ba["pos"].astype(np.float64) - bb["pos"].astype(np.float64), axis=1)
    dv = np.linalg.norm(ba["linVel"].astype(np.float64) - bb["linVel"].astype(np.float64), axis=1)
    return float(dp.max()), float(rot_angle(ba["quat"], bb["quat"]).max()), float(dv.max())


def fast_world(scene, **opts):
    g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1)
    for k, v in opts.items():
        g.w.set_option(k, v)
    return g


out = {}

# (i-a) reference snapshots
for name, scene in (("snapshot_Packed4_step25.npz", "Packed:4"), ("snapshot_Pyramid_step60.npz", "Pyramid"), ("snapshot_TwoBlockDrop_step120.npz", "TwoBlockDrop")):
    z = np.load(os.path.join(G, name), allow_pickle=False)
    g = fast_world(scene)
    g.set_bodies(z["bodies_before"].view(H.BODY_DT)); g.set_forces(z["forces_before"].view(H.FORCE_DT))
    if "step_index" in z.files:
        g.set_step_index(int(z["step_index"]))
    g.step()
    dp, dr, dv = body_diff(z["bodies_after"].view(H.BODY_DT), g.bodies())
    fa, fb = z["forces_after"].view(H.FORCE_DT), g.forces()
    same_forces = len(fa) == len(fb) and bool(np.array_equal(fa["bodyA"], fb["bodyA"]) and np.array_equal(fa["bodyB"], fb["bodyB"]) and np.array_equal(fa["numContacts"], fb["numContacts"]))
    dl = float(np.abs(fa["lambda"].astype(np.float64) - fb["lambda"].astype(np.float64)).max()) if same_forces and len(fa) else float("nan")
    lmax = float(np.abs(fa["lambda"]).max()) if len(fa) else 0.0
    out[name] = dict(scene=scene, dpos=dp, drot=dr, dvel=dv, same_force_list=same_forces, dlambda=dl, lambda_max=lmax, schedule=int(g.w.stats()["schedule"]))
    print(name, out[name]); sys.stdout.flush()
    g.close()

# (i-b) the reference's Stress1000 snapshot at step 140 (mid-landing) and the exact mode's own states: against the exact mode (bit-identical to the reference)
cases = [("Stress1000", 140), ("Stress1000", 400)] + ([("StressN:20", 410), ("StressN:40", 470)] if "--big" in sys.argv else [])
for scene, at in cases:
    ex = H.HipWorld(); ex.scene(scene)
    if at == 140:
        z = np.load(os.path.join(G, "stress1000_snapshot_step140.npz"), allow_pickle=False)
        b, f = z["bodies"].view(H.BODY_DT), z["forces"].view(H.FORCE_DT)
    else:
        ex.step(at)
        b, f = ex.bodies(), ex.forces()
    fa = fast_world(scene); fa.set_bodies(b); fa.set_forces(f); fa.set_step_index(at)
    e2 = H.HipWorld(); e2.scene(scene); e2.set_bodies(b); e2.set_forces(f); e2.set_step_index(at)
    rows = []
    for s in range(10):
        fa.step(); e2.step()
        rows.append(body_diff(e2.bodies(), fa.bodies()))
    out[f"{scene}@{at}"] = dict(step1=rows[0], step10=rows[9], plan=fa.w.fast_plan_info())
    print(scene, at, out[f"{scene}@{at}"]); sys.stdout.flush()
    ex.close(); fa.close(); e2.close()

# (ii) Stress1000, 600 steps against the reference's diagnostics; (iii) twice
diags = np.load(os.path.join(G, "stress1000_diag_600.npy")).view(H.DIAG_DT)
runs = []
for rep in range(2):
    g = fast_world("Stress1000")
    rec = []
    for s in range(600):
        g.step()
        d = g.diag()
        rec.append((int(d["activeContacts"]), int(d["activeManifolds"]), float(d["maxLinearSpeed"]), float(d["maxAngularSpeed"]), float(d["maxNormalImpulse"]), float(d["maxPenetration"])))
    b = g.bodies()
    runs.append((np.array(rec), b.copy()))
    g.close()
rec, b = runs[0]
dyn = b["invMass"] > 0
rc = rec[:, 0] / np.maximum(1, diags["activeContacts"]); rm = rec[:, 1] / np.maximum(1, diags["activeManifolds"])
v = b["linVel"][dyn].astype(np.float64); m = b["mass"][dyn].astype(np.float64)
out["stress1000_600"] = dict(
    first_contact_step=int(np.nonzero(rec[:, 0] > 0)[0][0]) + 1, ref_first_contact_step=int(np.nonzero(diags["activeContacts"] > 0)[0][0]) + 1,
    contacts_ratio_300_600=(float(rc[300:].min()), float(rc[300:].max())), manifolds_ratio_300_600=(float(rm[300:].min()), float(rm[300:].max())),
    contacts_600=(int(rec[599, 0]), int(diags["activeContacts"][599])), manifolds_600=(int(rec[599, 1]), int(diags["activeManifolds"][599])),
    max_lin_speed_500_600=(float(rec[500:, 2].max()), float(diags["maxLinearSpeed"][500:].max())),
    max_ang_speed_500_600=(float(rec[500:, 3].max()), float(diags["maxAngularSpeed"][500:].max())),
    max_impulse_500_600=(float(rec[500:, 4].max()), float(diags["maxNormalImpulse"][500:].max())),
    max_penetration=(float(rec[:, 5].max()), float(diags["maxPenetration"].max())),
    min_y=float(b["pos"][dyn][:, 1].min()), max_abs_xz=float(np.abs(b["pos"][dyn][:, [0, 2]]).max()), kinetic_energy=float(0.5 * np.sum(m * np.sum(v * v, axis=1))),
    run_to_run_bit_identical=bool(H.same_bits(runs[0][1], runs[1][1]) and np.array_equal(runs[0][0], runs[1][0])))
print("stress1000_600", out["stress1000_600"]); sys.stdout.flush()

# exact mode's end state for the same invariants
ex = H.HipWorld(); ex.scene("Stress1000"); ex.step(600)
be = ex.bodies(); ve = be["linVel"][dyn].astype(np.float64)
out["stress1000_600_exact"] = dict(min_y=float(be["pos"][dyn][:, 1].min()), max_abs_xz=float(np.abs(be["pos"][dyn][:, [0, 2]]).max()), kinetic_energy=float(0.5 * np.sum(m * np.sum(ve * ve, axis=1))))
print("stress1000_600_exact", out["stress1000_600_exact"])
ex.close()

for scene in ("Stack", "Pyramid"):
    g = fast_world(scene); g.step(300)
    e = H.HipWorld(); e.scene(scene); e.step(300)
    b, be = g.bodies(), e.bodies()
    dyn = b["invMass"] > 0
    out[scene + "_300"] = dict(max_abs_xz_drift_vs_exact=float(np.abs(b["pos"][dyn][:, [0, 2]] - be["pos"][dyn][:, [0, 2]]).max()), max_dy_vs_exact=float(np.abs(b["pos"][dyn][:, 1] - be["pos"][dyn][:, 1]).max()),
                               max_abs_x=float(np.abs(b["pos"][dyn][:, 0]).max()) if scene == "Stack" else None, contacts=(int(g.diag()["activeContacts"]), int(e.diag()["activeContacts"])))
    print(scene, out[scene + "_300"])
    g.close(); e.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "fast_tolerance.json"), "w"), indent=1)
