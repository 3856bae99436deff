"""HIP path vs the committed golden vectors of the real reference (no oracle, no /root/reference needed)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import harness as H
from golden.make_golden import joint_scene, state13

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(G, "golden_meta.json")))


def npz(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_stress1000_states_and_diagnostics_600_steps():
    """configs[1] end to end: every checkpoint state and all 600 diagnostics records of the reference run."""
    st = npz("stress1000_states.npz")
    diags = np.load(os.path.join(G, "stress1000_diag_600.npy")).view(H.DIAG_DT)
    g = H.HipWorld()
    g.scene("Stress1000")
    for s in range(600):
        g.step()
        d = np.array([g.diag()]).view(H.DIAG_DT)
        assert H.same_bits(d, diags[s:s + 1]), f"diagnostics of step {s + 1}: {d} vs {diags[s]}"
        key = f"step{s + 1}"
        if key in st.files:
            assert H.same_bits(state13(g.bodies()), st[key]), key


@pytest.mark.parametrize("name,scene", [("snapshot_Packed4_step25.npz", "Packed:4"), ("snapshot_Pyramid_step60.npz", "Pyramid"),
                                        ("snapshot_TwoBlockDrop_step120.npz", "TwoBlockDrop")])
def test_teacher_forced_step_from_reference_snapshot(name, scene):
    z = npz(name)
    g = H.HipWorld()
    g.scene(scene)
    g.set_bodies(z["bodies_before"].view(H.BODY_DT))
    g.set_forces(z["forces_before"].view(H.FORCE_DT))
    g.step()

    class Ref:
        def bodies(self): return z["bodies_after"].view(H.BODY_DT)
        def forces(self): return z["forces_after"].view(H.FORCE_DT)
        def diag(self): return z["diag_after"].view(H.DIAG_DT)[0]
    H.assert_same_world(Ref(), g, f"{scene}: one step from the reference's snapshot")


def test_joint_mini_scene_bug_compatible_rows():
    z = npz("joint_mini_scene.npz")
    g = H.HipWorld()
    joint_scene(g)
    for s in range(1, 5):
        g.step()

        class Ref:
            def bodies(self): return z[f"bodies{s}"].view(H.BODY_DT)
            def forces(self): return z[f"forces{s}"].view(H.FORCE_DT)
            def diag(self): return g.diag()
        H.assert_same_world(Ref(), g, f"joint mini scene step {s}", check_diag=False)


def test_kats_on_device():
    z = npz("kat_vectors.npz")
    g = H.HipWorld()
    out, cnt = g.w.kat_collide(z["posA"], z["quatA"], z["sizeA"], z["posB"], z["quatB"], z["sizeB"])
    assert np.array_equal(cnt, z["collide_counts"])
    for k in range(len(cnt)):
        got = out[k][:cnt[k]].view(H.CONTACT_DT)
        want = z["collide_contacts"][k][:cnt[k]].view(H.CONTACT_DT)
        for f in ("feature", "rA", "rB", "normal", "penetration"):
            assert H.same_bits(got[f], want[f]), (k, f)
    assert H.same_bits(g.w.kat_solve6(z["solve6_A"], z["solve6_b"]), z["solve6_x"])


def test_cli_trace_bytes():
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    for scene in ("Stack", "TwoBlockDrop"):
        want = gzip.open(os.path.join(G, f"trace_{scene}_300.txt.gz"), "rb").read()
        got = subprocess.run([exe, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
        assert got == want, scene


def test_pick_vectors_on_device():
    """avbd_pick (Solver::pick, solver.cpp:145-228) against the reference's answers on 400 rays, one launch."""
    z = npz("pick_vectors.npz")
    g = H.HipWorld()
    g.scene("Pyramid")
    g.step(60)
    body, local = g.w.pick(z["origins"], z["directions"])
    assert np.array_equal(body, z["body"])
    assert H.same_bits(local, z["local"])
