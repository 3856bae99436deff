"""Pins the CPU restatement (oracle/avbd_oracle.cpp) against the real reference build oracle/_ref.

Runs wherever oracle/_ref/libavbd_ref.so exists (built here from /root/reference by oracle/Makefile;
travels prebuilt to the GPU box).  Every comparison is bit-exact: full body state and the full force
list (order, contacts, lambda, penalty, C, fmin, fmax, stick flags).
"""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.skipif(not H.have("ref"), reason="oracle/_ref not built (reference sources absent)")


assert_same_world = H.assert_same_world


@pytest.mark.parametrize("scene,steps", [("Stack", 120), ("Pyramid", 60), ("Wall", 60), ("TwoBlockDrop", 200),
                                         ("Ground", 3), ("Empty", 2), ("Rod (WIP)", 20), ("Soft Body (WIP)", 80),
                                         ("StressN:3", 250), ("Packed:3", 60)])
def test_reference_scenes_full_state(scene, steps):
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    a.scene(scene); b.scene(scene)
    assert_same_world(a, b, f"{scene} step 0")
    for s in range(steps):
        a.step(); b.step()
        if s % 7 == 0 or s == steps - 1:
            assert_same_world(a, b, f"{scene} after step {s + 1}")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_piles_full_state(seed):
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    for w in (a, b):
        H.random_pile(w, np.random.default_rng(seed), 24)
        w.set_params(iterations=6 + seed)
    for s in range(90):
        a.step(); b.step()
        if s % 5 == 0:
            assert_same_world(a, b, f"pile seed {seed} step {s + 1}")
    assert a.diag()["activeContacts"] > 0


def test_post_stabilize_and_params():
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    for w in (a, b):
        H.random_pile(w, np.random.default_rng(11), 12)
        w.set_params(postStabilize=1, iterations=5, beta=5e4, gamma=0.98, alpha=0.9, gravity=(0.5, -9.0, 0.25))
    for s in range(60):
        a.step(); b.step()
    assert_same_world(a, b, "postStabilize")


def test_joints_springs_ignore_full_state():
    """Bug-compatible rows of joint.cpp / spring.cpp / ignorecollision.h."""
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    for w in (a, b):
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
    for s in range(12):
        a.step(); b.step()
        assert_same_world(a, b, f"joints step {s + 1}")


def test_collide_kat_random_pairs():
    rng = np.random.default_rng(5)
    hits = 0
    for k in range(1500):
        sA = rng.uniform(0.3, 1.5, 3).astype(np.float32)
        sB = rng.uniform(0.3, 1.5, 3).astype(np.float32)
        pA = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        pB = (pA + d * rng.uniform(0.2, 1.6)).astype(np.float32)
        qA = H.rand_quat(rng) if k % 3 else np.array([0, 0, 0, 1], np.float32)
        qB = H.rand_quat(rng) if k % 5 else np.array([0, 0, 0, 1], np.float32)
        ca = H.collide("orc", pA, qA, sA, pB, qB, sB)
        cb = H.collide("ref", pA, qA, sA, pB, qB, sB)
        assert len(ca) == len(cb)
        assert H.same_bits(ca, cb)
        hits += len(ca) > 0
    assert hits > 200


def test_solve_kats():
    rng = np.random.default_rng(7)
    for k in range(500):
        J = rng.normal(size=(6, 9)).astype(np.float32)
        M = (J @ J.T + np.eye(6, dtype=np.float32) * np.float32(rng.uniform(0.0, 3.0))).astype(np.float32)
        if k % 50 == 0:
            M[0, 0] = 0.0  # pivot guard path (maths.h:104)
        # blocks column-major: ll, la, al, aa
        A = np.concatenate([M[:3, :3].T.ravel(), M[:3, 3:].T.ravel(), M[3:, :3].T.ravel(), M[3:, 3:].T.ravel()]).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        assert H.same_bits(H.solve6("orc", A, b), H.solve6("ref", A, b))
        assert H.same_bits(H.solve3("orc", A[:9], b[:3]), H.solve3("ref", A[:9], b[:3]))
        q = H.rand_quat(rng)
        Ia, Ja = H.world_inertia("orc", (1.0, 0.5, 2.0), 1.5, q)
        Ib, Jb = H.world_inertia("ref", (1.0, 0.5, 2.0), 1.5, q)
        assert H.same_bits(Ia, Ib) and H.same_bits(Ja, Jb)


def test_pick_ties_inside_and_static_bodies():
    """Solver::pick: equal hit distances go to the newer body, a ray that starts inside a box picks it at t = 0, static
    bodies are never picked."""
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    for w in (a, b):
        w.add_body((30, 1, 30), 0.0, 0.5, (0, -0.5, 0))
        w.add_body((1, 1, 1), 1.0, 0.5, (0, 2.0, 0))
        w.add_body((1, 1, 1), 1.0, 0.5, (0, 2.0, 0))          # same box again: every hit is a tie
        w.add_body((1, 2, 1), 1.0, 0.5, (3, 2.0, 0), (0, 0, 0.3826834, 0.9238795))
    rays = [((0, 10, 0), (0, -1, 0)), ((0, 2.1, 0.2), (1, 0, 0)), ((10, 2, 0), (-3, 0, 0)), ((0, 10, 5), (0, -1, 0)),
            ((-10, 2, 0), (1, 0, 0)), ((3, 9, 0), (0, -2, 0)), ((0, 2, 0), (0, 0, 0)), ((0, 2, 0), (1e-4, 0, 0))]
    seen = []
    for o, d in rays:
        ra, la = a.pick(o, d); rb, lb = b.pick(o, d)
        assert ra == rb, (o, d)
        assert H.same_bits(la, lb), (o, d)
        seen.append(ra)
    assert seen[0] == 2 and seen[1] == 2 and seen[2] == 3 and seen[3] == -1 and seen[4] == 2 and seen[6] == -1 and seen[7] == -1


@pytest.mark.parametrize("seed", [5, 6])
def test_pick_random_pile(seed):
    a, b = H.CpuWorld("orc"), H.CpuWorld("ref")
    rng = np.random.default_rng(seed)
    for w in (a, b):
        H.random_pile(w, np.random.default_rng(seed), 30)
    a.step(40); b.step(40)
    hits = 0
    for k in range(300):
        o = rng.uniform(-8, 8, 3) + np.array([0, 8, 0]); t = rng.uniform(-3, 3, 3) + np.array([0, 1, 0])
        ra, la = a.pick(o, t - o); rb, lb = b.pick(o, t - o)
        assert ra == rb and H.same_bits(la, lb), k
        hits += ra >= 0
    assert hits > 30
