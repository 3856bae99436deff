"""Edge cases of the boundary on the device: empty / tiny scenes, bodies and forces added between steps, clear() and
rebuild, static-only scenes, a body with more than 64 incidence entries (generic path of k_iterate), both schedules."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def pair():
    return H.CpuWorld("orc"), H.HipWorld()


def test_empty_and_static_only_scenes():
    o, g = pair()
    for w in (o, g):
        w.scene("Empty")
    o.step(3); g.step(3)
    H.assert_same_world(o, g, "Empty")
    o, g = pair()
    for w in (o, g):
        w.add_body((4, 1, 4), 0.0, 0.5, (0, 0, 0))
        w.add_body((1, 1, 1), 0.0, 0.5, (0, 0.9, 0))      # two overlapping STATIC boxes: manifold with no swept endpoint
        w.add_body((1, 1, 1), 0.0, 0.5, (10, 0, 0))
    for s in range(5):
        o.step(); g.step()
        H.assert_same_world(o, g, f"static-static step {s + 1}")
    assert g.diag()["activeManifolds"] == 1 and g.diag()["dynamicBodies"] == 0


def test_single_free_body_and_zero_iterations():
    o, g = pair()
    for w in (o, g):
        w.add_body((1, 2, 0.5), 3.0, 0.5, (0, 5, 0), (0.1, 0.2, 0.3, 0.9273618), (1, 0, 0.5), (0.2, 3.0, -1.0))
        w.set_params(iterations=0)
    o.step(4); g.step(4)
    H.assert_same_world(o, g, "iterations=0")
    for w in (o, g):
        w.set_params(iterations=7)
    o.step(30); g.step(30)
    H.assert_same_world(o, g, "free body")


def test_bodies_and_forces_added_between_steps():
    o, g = pair()
    rng = np.random.default_rng(21)
    for w in (o, g):
        H.random_pile(w, np.random.default_rng(21), 10)
    o.step(15); g.step(15)
    for w in (o, g):
        w.add_body((1, 1, 1), 1.0, 0.5, (0.2, 7.0, 0.1))
        w.add_body((0.6, 0.6, 0.6), 2.0, 0.7, (0.0, 8.5, 0.0), (0, 0, 0.3826834, 0.9238795))
    o.step(10); g.step(10)
    H.assert_same_world(o, g, "after adding bodies at step 15")
    n = len(o.bodies())
    for w in (o, g):
        w.add_joint(n - 1, n - 2, (0, -0.3, 0), (0, 0.5, 0))
        w.add_ignore(1, 2)
    o.step(6); g.step(6)
    H.assert_same_world(o, g, "after adding a joint and an ignore marker at step 25")


def test_clear_and_rebuild_same_context():
    g = H.HipWorld()
    o = H.CpuWorld("orc")
    g.scene("Pyramid"); o.scene("Pyramid")
    g.step(20); o.step(20)
    g.scene("TwoBlockDrop"); o.scene("TwoBlockDrop")      # every scene function clears first (scenes.h)
    for s in range(30):
        g.step(); o.step()
    H.assert_same_world(o, g, "second scene in the same context")


def test_body_with_more_than_64_contact_entries_uses_generic_path():
    """A wide dynamic plate carrying 25 cubes: > 64 incidence entries on one body."""
    o, g = pair()
    for w in (o, g):
        w.add_body((40, 1, 40), 0.0, 0.5, (0, -0.5, 0))
        w.add_body((8, 0.5, 8), 0.5, 0.6, (0, 0.27, 0))
        for i in range(5):
            for j in range(5):
                w.add_body((1, 1, 1), 1.0, 0.5, ((i - 2) * 1.3, 1.04, (j - 2) * 1.3))
        w.set_params(iterations=8)
    most = 0
    for s in range(25):
        o.step(); g.step()
        H.assert_same_world(o, g, f"plate step {s + 1}")
    f = o.forces()
    plate_contacts = sum(int(r["numContacts"]) for r in f if 1 in (int(r["bodyA"]), int(r["bodyB"])))
    assert plate_contacts > 64, plate_contacts


def test_throughput_mode_and_level_mode_agree(monkeypatch):
    o = H.CpuWorld("orc"); o.scene("Packed:6"); o.step(30)
    for env in ({"AVBD_ITER_MODE": "1"}, {"AVBD_ITER_MODE": "2"}, {"AVBD_MODE": "levels"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = H.HipWorld(); g.scene("Packed:6"); g.step(30)
        for k in env:
            monkeypatch.delenv(k)
        H.assert_same_world(o, g, f"Packed:6 with {env}")
