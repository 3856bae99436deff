"""Pins the CPU restatement (oracle/) against the committed golden vectors of the REAL reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py from oracle/_ref (the
unmodified reference sources compiled where they lie).  This file needs neither /root/reference nor
oracle/_ref, so it also runs on the GPU box.  All comparisons are bit-exact.
"""
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import harness as H
from golden.make_golden import joint_scene, state13

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(G, "golden_meta.json")))
ORC_CLI = os.path.join(H.ORACLE_DIR, "oracle_cli")


def npz(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("scene", ["Empty", "Ground", "Stack", "Pyramid", "Wall", "TwoBlockDrop", "Rod (WIP)", "Soft Body (WIP)"])
def test_headless_trace_md5(scene):
    out = subprocess.run([ORC_CLI, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
    assert hashlib.md5(out).hexdigest() == META["trace_md5_300_steps"][scene]


@pytest.mark.parametrize("scene", ["Stack", "TwoBlockDrop"])
def test_headless_trace_bytes(scene):
    want = gzip.open(os.path.join(G, f"trace_{scene}_300.txt.gz"), "rb").read()
    got = subprocess.run([ORC_CLI, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
    assert got == want


def test_stress1000_checkpoints_and_diagnostics():
    st = npz("stress1000_states.npz")
    diags = np.load(os.path.join(G, "stress1000_diag_600.npy"))
    w = H.CpuWorld("orc")
    w.scene("Stress1000")
    for s in range(300):
        w.step()
        d = np.array([w.diag()])
        assert H.same_bits(d, diags[s:s + 1].view(H.DIAG_DT)), f"diagnostics of step {s + 1}"
        key = f"step{s + 1}"
        if key in st.files:
            assert H.same_bits(state13(w.bodies()), st[key]), key
    assert META["stress1000"]["first_contact_step"] == 132


@pytest.mark.slow
def test_stress1000_trace_md5_300_steps():
    out = subprocess.run([ORC_CLI, "--nogfx", "--scene", "Stress1000", "--steps", "300"], capture_output=True, check=True).stdout
    assert hashlib.md5(out).hexdigest() == META["trace_md5_300_steps"]["Stress1000"]


@pytest.mark.parametrize("name,scene", [("snapshot_Packed4_step25.npz", "Packed:4"), ("snapshot_Pyramid_step60.npz", "Pyramid"),
                                        ("snapshot_TwoBlockDrop_step120.npz", "TwoBlockDrop")])
def test_teacher_forced_step_from_reference_snapshot(name, scene):
    z = npz(name)
    w = H.CpuWorld("orc")
    w.scene(scene)
    w.set_bodies(z["bodies_before"].view(H.BODY_DT))
    w.set_forces(z["forces_before"].view(H.FORCE_DT))
    w.step()

    class Ref:
        def bodies(self): return z["bodies_after"].view(H.BODY_DT)
        def forces(self): return z["forces_after"].view(H.FORCE_DT)
        def diag(self): return z["diag_after"].view(H.DIAG_DT)[0]
    H.assert_same_world(Ref(), w, f"{scene}: one step from the reference's snapshot")


def test_joint_mini_scene():
    z = npz("joint_mini_scene.npz")
    w = H.CpuWorld("orc")
    joint_scene(w)
    for s in range(0, 5):
        if s:
            w.step()

        class Ref:
            def bodies(self): return z[f"bodies{s}"].view(H.BODY_DT)
            def forces(self): return z[f"forces{s}"].view(H.FORCE_DT)
            def diag(self): return w.diag()
        H.assert_same_world(Ref(), w, f"joint mini scene step {s}", check_diag=False)


def test_collide_and_solve_kats():
    z = npz("kat_vectors.npz")
    for k in range(len(z["posA"])):
        c = H.collide("orc", z["posA"][k], z["quatA"][k], z["sizeA"][k], z["posB"][k], z["quatB"][k], z["sizeB"][k])
        assert len(c) == z["collide_counts"][k]
        assert H.same_bits(c, z["collide_contacts"][k][:len(c)].view(H.CONTACT_DT))
    for k in range(len(z["solve6_A"])):
        assert H.same_bits(H.solve6("orc", z["solve6_A"][k], z["solve6_b"][k]), z["solve6_x"][k])


def test_pick_vectors():
    """Solver::pick (solver.cpp:145-228) of the restatement against the reference's answers on 400 rays."""
    z = npz("pick_vectors.npz")
    w = H.CpuWorld("orc")
    w.scene("Pyramid")
    w.step(60)
    assert (z["body"] >= 0).sum() > 200 and (z["body"] < 0).sum() > 50
    for k in range(len(z["body"])):
        b, local = w.pick(z["origins"][k], z["directions"][k])
        assert b == z["body"][k], k
        assert H.same_bits(local, z["local"][k]), k
