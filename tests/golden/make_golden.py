#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference build (oracle/_ref).

Run in the build container (where /root/reference exists and `make -C oracle ref` has produced
oracle/_ref/libavbd_ref.so + avbd_ref):

    python tests/golden/make_golden.py

Fixtures are DATA ONLY (inputs and expected outputs of the unmodified reference sources); no reference
source text is stored.  They let tests/test_oracle_golden.py pin the CPU restatement, and
tests/test_gpu_golden.py pin the HIP path, on machines where neither /root/reference nor oracle/_ref exist.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402
from harness import build_row_api_scene  # noqa: E402

REF_CLI = os.path.join(H.ORACLE_DIR, "_ref", "avbd_ref")


def state13(b):
    return np.concatenate([b["pos"], b["quat"], b["linVel"], b["angVel"]], axis=1).astype(np.float32)


def main():
    assert H.have("ref") and os.path.exists(REF_CLI), "build oracle/_ref first (make -C oracle ref)"
    meta = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref (unmodified /root/reference/source, g++ -O2 -ffp-contract=off)"}

    # 1. headless traces: md5 of the full stdout for every reference scene, 300 steps
    md5 = {}
    for scene in ["Empty", "Ground", "Stack", "Pyramid", "Wall", "TwoBlockDrop", "Stress1000", "Rod (WIP)", "Soft Body (WIP)"]:
        out = subprocess.run([REF_CLI, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
        md5[scene] = hashlib.md5(out).hexdigest()
        if scene in ("Stack", "TwoBlockDrop"):
            with gzip.open(os.path.join(HERE, f"trace_{scene}_300.txt.gz"), "wb", compresslevel=9) as f:
                f.write(out)
    meta["trace_md5_300_steps"] = md5

    # 1b. user-defined Force subclasses (tests/host/user_forces.h) compiled against the reference's own headers and sources:
    #     the floats the reference computes for a user's Tether / Pin (oracle/user_force_ref.cpp, built by `make -C oracle ref`)
    uf = os.path.join(H.ORACLE_DIR, "_ref", "user_force_ref")
    subprocess.run([uf, os.path.join(HERE, "user_force_states.bin")], check=True)
    import hashlib
    meta["user_force_states.bin"] = {"what": "tests/host/user_forces.h (Tether, Pin, Fuse) compiled against /root/reference/source by oracle/user_force_ref.cpp: "
                                             "body states every 10 steps + the user force's rows",
                                     "md5": hashlib.md5(open(os.path.join(HERE, "user_force_states.bin"), "rb").read()).hexdigest()}

    # 2. Stress1000: 13-float body state at the survey's checkpoints + the per-step diagnostics for 600 steps
    w = H.CpuWorld("ref")
    w.scene("Stress1000")
    diags = np.zeros(600, H.DIAG_DT)
    states = {}
    for s in range(600):
        w.step()
        diags[s] = w.diag()
        if s + 1 in (1, 131, 132, 140, 200, 300, 400, 600):
            states[f"step{s + 1}"] = state13(w.bodies())
        if s + 1 == 140:
            np.savez_compressed(os.path.join(HERE, "stress1000_snapshot_step140.npz"), bodies=w.bodies(), forces=w.forces())
    np.savez_compressed(os.path.join(HERE, "stress1000_states.npz"), **states)
    # 2b. row API through the reference's own virtuals (solver.h:102-108): Force::computeConstraint(alpha) on every force of a
    #     snapshot (with its in-place mutations), then computeDerivatives for both endpoints and every row
    rows = {}
    for tag, scene, steps, alpha in (("pyramid", "Pyramid", 40, 0.95), ("jointed", "@jointed", 2, 0.3), ("stack", "Stack", 25, 1.0)):
        r = H.CpuWorld("ref"); build_row_api_scene(r, scene); r.step(steps)
        before_b, before_f = r.bodies(), r.forces()
        nf = len(before_f)
        derivs = np.zeros((nf, 2, 12, 6), np.float32)
        for i in range(nf):
            nrow = r.force_compute_constraint(i, alpha)
            assert nrow == int(before_f["rows"][i])
            for e, body in enumerate((int(before_f["bodyA"][i]), int(before_f["bodyB"][i]))):
                for row in range(nrow):
                    jl, ja = r.force_compute_derivatives(i, body, row)
                    derivs[i, e, row, :3] = jl; derivs[i, e, row, 3:] = ja
        rows[f"{tag}_bodies"] = before_b; rows[f"{tag}_forces"] = before_f; rows[f"{tag}_after"] = r.forces()
        rows[f"{tag}_derivs"] = derivs; rows[f"{tag}_alpha"] = np.float32(alpha); rows[f"{tag}_steps"] = np.int32(steps)
    np.savez_compressed(os.path.join(HERE, "row_api_vectors.npz"), **rows)

    np.save(os.path.join(HERE, "stress1000_diag_600.npy"), diags)
    meta["stress1000"] = {"first_contact_step": int(np.argmax(diags["activeContacts"] > 0)) + 1,
                          "step600": {k: (float(diags[599][k]) if diags[599][k].dtype.kind == "f" else int(diags[599][k])) for k in H.DIAG_DT.names}}

    # 3. full snapshots (bodies + force list newest-first) for teacher-forced single-step tests
    for scene, steps in (("Packed:4", 25), ("Pyramid", 60), ("TwoBlockDrop", 120)):
        w = H.CpuWorld("ref")
        w.scene(scene)
        w.step(steps)
        b0, f0 = w.bodies(), w.forces()
        w.step()
        b1, f1 = w.bodies(), w.forces()
        name = scene.replace(":", "")
        np.savez_compressed(os.path.join(HERE, f"snapshot_{name}_step{steps}.npz"), bodies_before=b0, forces_before=f0,
                            bodies_after=b1, forces_after=f1, diag_after=np.array([w.diag()]))

    # 4. joints / springs / ignore mini-scene (bug-compatible rows): states for steps 0..4
    w = H.CpuWorld("ref")
    joint_scene(w)
    js = {"bodies0": w.bodies(), "forces0": w.forces()}
    for s in range(1, 5):
        w.step()
        js[f"bodies{s}"] = w.bodies()
        js[f"forces{s}"] = w.forces()
    np.savez_compressed(os.path.join(HERE, "joint_mini_scene.npz"), **js)

    # 5. known-answer vectors
    rng = np.random.default_rng(2026)
    n = 400
    sA = rng.uniform(0.3, 1.5, (n, 3)).astype(np.float32); sB = rng.uniform(0.3, 1.5, (n, 3)).astype(np.float32)
    pA = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pB = (pA + d * rng.uniform(0.2, 1.6, (n, 1))).astype(np.float32)
    qA = np.stack([H.rand_quat(rng) if k % 3 else np.array([0, 0, 0, 1], np.float32) for k in range(n)])
    qB = np.stack([H.rand_quat(rng) if k % 5 else np.array([0, 0, 0, 1], np.float32) for k in range(n)])
    counts = np.zeros(n, np.int32); contacts = np.zeros((n, 4), H.CONTACT_DT)
    for k in range(n):
        c = H.collide("ref", pA[k], qA[k], sA[k], pB[k], qB[k], sB[k])
        counts[k] = len(c); contacts[k, :len(c)] = c
    As, bs, xs = [], [], []
    for k in range(300):
        J = rng.normal(size=(6, 9)).astype(np.float32)
        M = (J @ J.T + np.eye(6, dtype=np.float32) * np.float32(rng.uniform(0.0, 3.0))).astype(np.float32)
        if k % 50 == 0:
            M[0, 0] = 0.0
        A = np.concatenate([M[:3, :3].T.ravel(), M[:3, 3:].T.ravel(), M[3:, :3].T.ravel(), M[3:, 3:].T.ravel()]).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        As.append(A); bs.append(b); xs.append(H.solve6("ref", A, b))
    np.savez_compressed(os.path.join(HERE, "kat_vectors.npz"), posA=pA, quatA=qA, sizeA=sA, posB=pB, quatB=qB, sizeB=sB,
                        collide_counts=counts, collide_contacts=contacts, solve6_A=np.array(As), solve6_b=np.array(bs), solve6_x=np.array(xs))

    pick_vectors()

    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        print(f"  {fn:45s} {os.path.getsize(os.path.join(HERE, fn)):>9d} B")


def pick_rays(rng, n):
    """Rays at the Pyramid scene: mostly aimed at the pile from outside, plus axis-parallel, inside-a-box, unnormalised,
    degenerate and missing ones (same generator for the fixture and for live comparisons)."""
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    for k in range(n):
        target = np.array([rng.uniform(-6, 6), rng.uniform(0, 6), rng.uniform(-1, 1)])
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        origin = target + u * rng.uniform(8, 20)
        direction = (target - origin) * rng.uniform(0.01, 30.0)
        m = k % 10
        if m == 0:      # axis-parallel: exercises the |d| < epsilon slab branch
            axis = k // 10 % 3
            direction = np.zeros(3); direction[axis] = -1.0 if origin[axis] > target[axis] else 1.0
            origin = target.copy(); origin[axis] -= direction[axis] * 15.0
        elif m == 1:    # starts inside the pile
            origin = np.array([rng.uniform(-4, 4), rng.uniform(0.1, 3), rng.uniform(-0.4, 0.4)])
        elif m == 2 and k % 20 == 2:
            direction = direction * 1e-6    # shorter than sqrt(1e-6): no pick
        elif m == 3 and k % 20 == 3:
            direction = -direction          # points away
        o[k] = origin; d[k] = direction
    return o, d


def pick_vectors():
    """Solver::pick of the unmodified reference on the Pyramid scene after 60 steps (the state of snapshot_Pyramid_step60)."""
    w = H.CpuWorld("ref")
    w.scene("Pyramid")
    w.step(60)
    o, d = pick_rays(np.random.default_rng(77), 400)
    body = np.zeros(len(o), np.int32); local = np.zeros((len(o), 3), np.float32)
    for k in range(len(o)):
        body[k], local[k] = w.pick(o[k], d[k])
    np.savez_compressed(os.path.join(HERE, "pick_vectors.npz"), origins=o, directions=d, body=body, local=local)
    return body


def joint_scene(w):
    """Ground + welded pair + world-anchored tilted box + two springs + an ignored pair (same calls on any backend)."""
    w.add_body((30, 1, 30), 0.0, 0.5, (0, -0.5, 0))
    i1 = w.add_body((1, 1, 1), 1.0, 0.5, (0, 3.0, 0))
    i2 = w.add_body((1, 1, 1), 1.0, 0.5, (1.0, 3.0, 0))
    i3 = w.add_body((1, 0.5, 1), 2.0, 0.4, (4.0, 2.0, 0), (0, 0, 0.1736482, 0.9848078))
    i4 = w.add_body((0.5, 0.5, 0.5), 1.0, 0.5, (4.0, 4.0, 0.3))
    i5 = w.add_body((1, 1, 1), 1.0, 0.5, (-4.0, 0.5, 0))
    i6 = w.add_body((1, 1, 1), 1.0, 0.5, (-4.0, 1.45, 0))
    w.add_joint(i1, i2, (0.5, 0, 0), (-0.5, 0, 0))
    w.add_joint(-1, i3, (4.0, 2.5, 0))
    w.add_spring(i3, i4, (0, 0.25, 0), (0, -0.25, 0), 500.0)
    w.add_spring(i1, i4, (0, 0, 0), (0, 0, 0), 200.0, 2.5)
    w.add_ignore(i5, i6)


if __name__ == "__main__":
    if sys.argv[1:] == ["pick"]:      # only the pick fixture (the others are unchanged by it)
        b = pick_vectors()
        print("pick_vectors.npz:", int((b >= 0).sum()), "hits of", len(b))
    else:
        main()
