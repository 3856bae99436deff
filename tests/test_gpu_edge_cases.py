"""Edge cases of the boundary on the device: empty / tiny scenes, bodies and forces added between steps, clear() and
rebuild, static-only scenes, a body with more than 64 incidence entries (generic path of k_iterate), both schedules."""
import os

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


def test_resident_and_task_schedules_are_both_exercised_and_agree(monkeypatch):
    """Contact-only scenes with few enough bodies run the resident schedule (stats.schedule == 2); AVBD_RESIDENT=0 forces
    the task schedule (1).  Both must reproduce the CPU oracle bit for bit, including post-stabilisation and a scene
    that changes between steps."""
    for scene, steps in (("Stack", 40), ("Pyramid", 25), ("Stress1000", 60)):
        o = H.CpuWorld("orc"); o.scene(scene); o.step(steps)
        for resident in ("1", "0"):
            monkeypatch.setenv("AVBD_RESIDENT", resident)
            g = H.HipWorld(); g.scene(scene); g.step(steps)
            monkeypatch.delenv("AVBD_RESIDENT")
            assert int(g.w.stats()["schedule"]) == (2 if resident == "1" else 1), (scene, resident, int(g.w.stats()["schedule"]))
            H.assert_same_world(o, g, f"{scene} resident={resident}")


def test_resident_schedule_post_stabilize_and_odd_iteration_counts():
    for iters, post in ((1, False), (3, True), (7, False), (20, True), (59, True)):
        o, g = pair()
        for w in (o, g):
            w.scene("Stack")
            w.set_params(iterations=iters, postStabilize=post)
        for s in range(12):
            o.step(); g.step()
        assert int(g.w.stats()["schedule"]) == 2
        H.assert_same_world(o, g, f"Stack iterations={iters} postStabilize={post}")


def test_resident_schedule_vetoed_on_device_falls_back_to_tasks():
    """> 64 entries on one swept body: the host cannot know, the device vetoes (Counters.residentBad) and the same launch
    runs the task schedule; a scene with a spring is not offered to the resident schedule at all (joints are)."""
    o, g = pair()
    for w in (o, g):
        w.add_body((40, 1, 40), 0.0, 0.5, (0, -0.5, 0))
        w.add_body((8, 0.5, 8), 0.5, 0.6, (0, 0.27, 0))
        for i in range(5):
            for j in range(5):
                w.add_body((1, 1, 1), 1.0, 0.5, ((i - 2) * 1.3, 1.6, (j - 2) * 1.3))   # dropped: the plate starts with 4 contacts
    seen = set()
    for s in range(35):
        o.step(); g.step()
        seen.add(int(g.w.stats()["schedule"]))
        H.assert_same_world(o, g, f"plate step {s + 1}")
    assert seen == {1, 2}, seen          # resident while the plate has few contacts, tasks once it carries > 64
    o, g = pair()
    for w in (o, g):
        w.scene("Stack")
        w.add_joint(3, 4, (0, 0.5, 0), (0, -0.5, 0))
    o.step(10); g.step(10)
    assert int(g.w.stats()["schedule"]) == 2                      # joints are taken by the resident schedule ...
    H.assert_same_world(o, g, "Stack + joint")
    o, g = pair()
    for w in (o, g):
        w.scene("Stack")
        w.add_spring(3, 5, (0, 0.5, 0), (0, -0.5, 0), 800.0)
    o.step(10); g.step(10)
    assert int(g.w.stats()["schedule"]) == 1                      # ... springs are not: the scene is not offered to it
    H.assert_same_world(o, g, "Stack + spring")


def test_too_many_iterations_for_the_stamp_window_uses_tasks():
    o, g = pair()
    for w in (o, g):
        w.scene("Stack")
        w.set_params(iterations=64)
    o.step(5); g.step(5)
    assert int(g.w.stats()["schedule"]) == 1
    H.assert_same_world(o, g, "Stack iterations=64")


def test_fused_topology_kernel_fast_and_slow_paths(monkeypatch):
    """k_topology_small stages at most 6144 adjacency entries in LDS; a packing of 3376 cubes has more and takes its
    in-kernel global-memory path (Packed:15: 3376 cubes, > 3072 manifolds); AVBD_FUSED_TOPOLOGY=0 is the multi-launch chain.  All three agree with the oracle."""
    for scene, steps in (("Packed:15", 12), ("Stress1000", 40)):
        o = H.CpuWorld("orc"); o.scene(scene); o.step(steps)
        for fused in ("1", "0"):
            monkeypatch.setenv("AVBD_FUSED_TOPOLOGY", fused)
            g = H.HipWorld(); g.scene(scene); g.step(steps)
            monkeypatch.delenv("AVBD_FUSED_TOPOLOGY")
            H.assert_same_world(o, g, f"{scene} fused_topology={fused}")
        if scene == "Packed:15":
            assert 2 * int(g.w.stats()["numManifolds"]) > 6144


def test_front_end_and_polling_knobs_do_not_change_results(monkeypatch):
    """AVBD_FUSED_FRONT=0 (separate broadphase / narrow-phase launches) and the quiet share of the resident schedule's wait
    (0 = poll at once, 900 = sleep through 90 % of the previous wait) are placement/timing choices only, and so is the sweep
    order handed to the data-flow schedules (descending creation index by default, dependency levels with AVBD_ORDER=levels)."""
    o = H.CpuWorld("orc"); o.scene("Pyramid"); o.step(40)
    for env in ({"AVBD_FUSED_FRONT": "0"}, {"AVBD_RESIDENT_QUIET": "0"}, {"AVBD_RESIDENT_QUIET": "900"}, {"AVBD_ORDER": "levels"},
                {"AVBD_ORDER": "levels", "AVBD_RESIDENT": "0"}, {"AVBD_ORDER": "levels", "AVBD_FUSED_TOPOLOGY": "0", "AVBD_RESIDENT": "0"},
                {"AVBD_FUSED_FRONT": "0", "AVBD_FUSED_TOPOLOGY": "0", "AVBD_RESIDENT": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = H.HipWorld(); g.scene("Pyramid"); g.step(40)
        for k in env:
            monkeypatch.delenv(k)
        H.assert_same_world(o, g, f"Pyramid with {env}")


def test_resident_stamp_window_wraps_cleanly(monkeypatch):
    """Stamps are launch serial * 64 + iteration; at serial 2^24 the exchange buffers are cleared and the serial restarts.
    Start 10 launches before the wrap and step across it."""
    monkeypatch.setenv("AVBD_RESIDENT_SERIAL", str((1 << 24) - 12))
    o, g = pair()
    monkeypatch.delenv("AVBD_RESIDENT_SERIAL")
    for w in (o, g):
        w.scene("Stack")
    for s in range(30):
        o.step(); g.step()
        H.assert_same_world(o, g, f"Stack across the stamp wrap, step {s + 1}")
    assert int(g.w.stats()["schedule"]) == 2


def test_resident_schedule_with_several_bodies_per_wave(monkeypatch):
    """More swept bodies than resident wavefronts (1728 > 1024): every wave takes every W-th body of the level order and
    parks a body's state between its tasks (stats.schedule == 3).  AVBD_RESIDENT_MULTI=0 sends the scene to the task
    schedule instead (1).  Post-stabilisation changes which task is a body's last."""
    monkeypatch.setenv("AVBD_GROUPS", "0")                       # (by default a scene of this size takes the group schedule, test_gpu_groups.py)
    for post in (0, 1):
        o = H.CpuWorld("orc"); o.scene("Packed:12"); o.set_params(postStabilize=post, iterations=6); o.step(14)
        for multi, sched in (("1", 3), ("0", 1)):
            monkeypatch.setenv("AVBD_RESIDENT_MULTI", multi)
            g = H.HipWorld(); g.scene("Packed:12"); g.set_params(postStabilize=post, iterations=6); g.step(14)
            monkeypatch.delenv("AVBD_RESIDENT_MULTI")
            assert int(g.w.stats()["schedule"]) == sched
            H.assert_same_world_fast(o, g, f"Packed:12 postStabilize={post} multi={multi}", check_rows=True)


def test_snapshot_save_load_continues_bit_for_bit(tmp_path):
    """SURVEY.md §8 f3 (checkpoint / resume, absent in the reference): a context that loads a snapshot goes on exactly like
    the one that saved it — contacts only (resident schedule) and with joints + springs (task schedule)."""
    for scene, pre, post in (("Pyramid", 45, 30), ("Soft Body (WIP)", 30, 25), ("Stress1000", 200, 40)):
        a = H.HipWorld(); a.scene(scene); a.step(pre)
        path = tmp_path / f"{scene.split()[0]}.snap"
        a.w.save_snapshot(path)
        b = H.HipWorld(); b.scene("Stack")                       # whatever was there is replaced
        b.w.load_snapshot(path)
        assert b.w.step_index() == a.w.step_index() == pre
        H.assert_same_world(a, b, f"{scene}: right after the load", check_rows=True, check_diag=False)
        a.step(post); b.step(post)
        H.assert_same_world(a, b, f"{scene}: {post} steps after the load")
    with pytest.raises(Exception):
        b.w.load_snapshot(tmp_path / "missing.snap")


def test_cli_save_state_load_state(tmp_path):
    import subprocess
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    snap = str(tmp_path / "stack.snap")
    full = subprocess.run([exe, "--nogfx", "--scene", "Stack", "--steps", "70"], capture_output=True, text=True, check=True).stdout.splitlines()
    subprocess.run([exe, "--nogfx", "--scene", "Stack", "--steps", "40", "--save-state", snap], capture_output=True, text=True, check=True)
    tail = subprocess.run([exe, "--nogfx", "--load-state", snap, "--steps", "30"], capture_output=True, text=True, check=True).stdout.splitlines()
    body = lambda lines: [l for l in lines if l.startswith("  Body") or l.startswith("  Diagnostics")]
    a, b = body(full), body(tail)
    n = len(b)
    assert n > 0 and a[-n:] == b


def test_non_finite_state_is_sanitised_like_the_reference():
    """solver.cpp:41-66: non-finite velocities / poses are reset (and counted) instead of spreading.  Inject them through the
    body upload on both sides and keep comparing bit for bit."""
    o, g = pair()
    for w in (o, g):
        w.scene("Pyramid")
    o.step(12); g.step(12)
    b = g.bodies().copy()
    b["linVel"][5] = (np.nan, 1.0, np.inf)
    b["angVel"][9] = (np.inf, 0.0, -np.inf)
    b["linVel"][20] = (-np.inf, np.nan, 0.0)
    f = o.forces()
    o.set_bodies(b); o.set_forces(f)
    g.set_bodies(b); g.set_forces(f)
    for s in range(6):
        o.step(); g.step()
        H.assert_same_world(o, g, f"after injecting non-finite velocities, step {s + 1}", check_diag=(s > 0))
    gb = g.bodies()
    assert np.isfinite(gb["pos"]).all() and np.isfinite(gb["linVel"]).all() and np.isfinite(gb["angVel"]).all()


def _pile(w, skip=None, joints=()):
    w.add_body((30, 1, 30), 0.0, 0.5, (0, -0.5, 0))
    k = 1
    for layer in range(3):
        for i in range(2):
            for j in range(2):
                if k != skip:
                    w.add_body((1, 1, 1), 1.0, 0.5, ((i - 0.5) * 1.02, 0.5 + layer * 1.001, (j - 0.5) * 1.02))
                k += 1
    for a, b in joints:
        w.add_joint(a, b, (0.5, 0, 0), (-0.5, 0, 0))
    w.set_params(iterations=8)


def test_remove_body_rebuilds_the_state_without_it():
    """`delete body` of the reference: avbd_remove_body drops the body and the forces attached to it and renumbers the later
    bodies.  Checked against the oracle started from the same filtered state in a scene built without that body."""
    g = H.HipWorld(); _pile(g); g.step(20)
    victim = 6
    b, f = g.bodies(), g.forces()
    g.w.remove_body(victim)
    assert g.w.num_bodies() == len(b) - 1
    keep = [i for i in range(len(f)) if victim not in (int(f["bodyA"][i]), int(f["bodyB"][i]))]
    f2 = f[keep].copy()
    for name in ("bodyA", "bodyB"):
        f2[name] = np.where(f2[name] > victim, f2[name] - 1, f2[name])
    b2 = np.delete(b, victim)
    o = H.CpuWorld("orc"); _pile(o, skip=victim)
    o.set_bodies(b2); o.set_forces(f2)
    # (initial / inertial poses are scratch of the predictor: rewritten at the start of every step, not part of set_bodies)
    for s in range(15):
        o.step(); g.step()
        H.assert_same_world(o, g, f"{s + 1} steps after the removal", check_diag=(s > 0))


def test_remove_force_drops_one_joint():
    g = H.HipWorld(); _pile(g, joints=((1, 2), (3, 4))); g.step(10)
    b, f = g.bodies(), g.forces()
    joint_rows = [i for i in range(len(f)) if int(f["kind"][i]) == 1 and {int(f["bodyA"][i]), int(f["bodyB"][i])} == {3, 4}]
    assert len(joint_rows) == 1
    g.w.remove_force(joint_rows[0])
    f2 = np.delete(f, joint_rows[0])
    o = H.CpuWorld("orc"); _pile(o, joints=((1, 2),))
    o.set_bodies(b); o.set_forces(f2)
    for s in range(10):
        o.step(); g.step()
        H.assert_same_world(o, g, f"{s + 1} steps after removing the joint", check_diag=(s > 0))


def test_cli_delete_body_through_the_host_mirror(tmp_path):
    """`delete body` in the source-compatible host API (a body that already lives on the device): same final state as
    avbd_remove_body through the C ABI at the same step."""
    import subprocess
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    dump = str(tmp_path / "final.bin")
    subprocess.run([exe, "--nogfx", "--scene", "Pyramid", "--steps", "40", "--remove-body", "17", "--remove-at", "25", "--quiet", "--dump", dump],
                   capture_output=True, text=True, check=True)
    got = np.fromfile(dump, dtype=H.load_product().BODY_DT)
    g = H.HipWorld(); g.scene("Pyramid"); g.step(25); g.w.remove_body(17); g.step(15)
    want = g.w.bodies()
    assert len(got) == len(want) == 55
    assert H.same_bits(got, want)


def test_pick_ties_static_bodies_and_empty_world():
    """avbd_pick: ties go to the newer body, static bodies and empty worlds give no pick, rays are batched."""
    o, g = pair()
    assert g.w.pick([(0, 5, 0)], [(0, -1, 0)])[0][0] == -1           # nothing in the world yet
    for w in (o, g):
        w.add_body((30, 1, 30), 0.0, 0.5, (0, -0.5, 0))
    assert g.w.pick([(0, 5, 0)], [(0, -1, 0)])[0][0] == -1           # only a static body
    for w in (o, g):
        w.add_body((1, 1, 1), 1.0, 0.5, (0, 2.0, 0))
        w.add_body((1, 1, 1), 1.0, 0.5, (0, 2.0, 0))
        w.add_body((1, 2, 1), 1.0, 0.5, (3, 2.0, 0), (0, 0, 0.3826834, 0.9238795))
    rays = [((0, 10, 0), (0, -1, 0)), ((0, 2.1, 0.2), (1, 0, 0)), ((10, 2, 0), (-3, 0, 0)), ((0, 10, 5), (0, -1, 0)),
            ((-10, 2, 0), (1, 0, 0)), ((3, 9, 0), (0, -2, 0)), ((0, 2, 0), (0, 0, 0)), ((0, 2, 0), (1e-4, 0, 0))]
    body, local = g.w.pick([r[0] for r in rays], [r[1] for r in rays])
    for k, (ro, rd) in enumerate(rays):
        b, l = o.pick(ro, rd)
        assert body[k] == b, k
        assert H.same_bits(local[k], l), k
    assert list(body[:5]) == [2, 2, 3, -1, 2]


def test_pick_after_stepping_a_large_pile_matches_the_oracle():
    o, g = pair()
    for w in (o, g):
        w.scene("Packed:6")
    o.step(15); g.step(15)
    rng = np.random.default_rng(3)
    origins = rng.uniform(-10, 10, (256, 3)).astype(np.float32) + np.float32([0, 12, 0])
    targets = rng.uniform(-3, 3, (256, 3)).astype(np.float32) + np.float32([0, 3, 0])
    body, local = g.w.pick(origins, targets - origins)
    assert (body >= 0).sum() > 100
    for k in range(256):
        b, l = o.pick(origins[k], targets[k] - origins[k])
        assert body[k] == b and H.same_bits(local[k], l), k


def test_cli_pick_through_the_host_mirror():
    """Solver::pick of the source-compatible host API: same body and bit-identical local point as the C ABI call."""
    import subprocess
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    out = subprocess.run([exe, "--nogfx", "--scene", "Pyramid", "--steps", "30", "--quiet", "--pick", "0.3,20,0.1,0.01,-1,0.02"],
                         capture_output=True, text=True, check=True).stdout
    line = [l for l in out.splitlines() if l.startswith("Pick:")][0]
    g = H.HipWorld(); g.scene("Pyramid"); g.step(30)
    body, local = g.w.pick([(0.3, 20, 0.1)], [(0.01, -1, 0.02)])
    assert body[0] >= 0
    want = "Pick: body id=%d index=%d local=(%s, %s, %s)" % (body[0], body[0], *[float(v).hex() for v in local[0]])
    import re
    m = re.match(r"Pick: body id=(\d+) index=(\d+) local=\((.*), (.*), (.*)\)", line)
    assert m, line
    assert int(m.group(2)) == body[0]
    got = np.float32([float.fromhex(m.group(k)) for k in (3, 4, 5)])
    assert H.same_bits(got, local[0]), (line, want)
    miss = subprocess.run([exe, "--nogfx", "--scene", "Pyramid", "--steps", "1", "--quiet", "--pick", "0,50,0,0,1,0"],
                          capture_output=True, text=True, check=True).stdout
    assert "Pick: none" in miss


@pytest.mark.parametrize("ooo", ["0", "1"])
def test_two_workgroups_per_cu_with_a_body_of_more_than_34_entries(monkeypatch, ooo):
    """The launch with two workgroups per CU keeps short product lines (rows 36..103); a body with more entries than that
    (a dynamic plate carrying 3 x 4 cubes: ~52 contacts) multiplies its last rows after the wait.  Few resident waves
    (AVBD_ITER_BLOCKS=2: 16 waves for 29 swept bodies) so that every wave owns several bodies, swept in order or out of order."""
    monkeypatch.setenv("AVBD_RESIDENT_FORCE_LITE", "1")
    monkeypatch.setenv("AVBD_ITER_BLOCKS", "2")
    monkeypatch.setenv("AVBD_RESIDENT_OOO", ooo)
    monkeypatch.setenv("AVBD_GROUPS", "0")                       # the resident schedule's parked sweep is what this test is about
    o, g = pair()
    for w in (o, g):
        w.add_body((40, 1, 40), 0.0, 0.5, (0, -0.5, 0))
        w.add_body((6, 0.5, 8), 0.5, 0.6, (0, 0.27, 0))
        for i in range(3):
            for j in range(4):
                w.add_body((1, 1, 1), 1.0, 0.5, ((i - 1) * 1.4, 1.04, (j - 1.5) * 1.4))
        for i in range(4):
            for j in range(4):
                w.add_body((1, 1, 1), 1.0, 0.5, (8 + i * 1.1, 0.52 + 1.03 * j, 0))
        w.set_params(iterations=7)
    for s in range(30):
        o.step(); g.step()
        H.assert_same_world(o, g, f"plate under the short product lines, step {s + 1}")
    assert int(g.w.stats()["schedule"]) == 3
    f = o.forces()
    plate = sum(int(r["numContacts"]) for r in f if 1 in (int(r["bodyA"]), int(r["bodyB"])))
    assert 34 < plate <= 64, plate


def _c_float3(v):
    """printf("%.3f") of glibc for one float, non-finite values included (Python prints no sign for NaN)."""
    v = float(v)
    if np.isnan(v):
        return "-nan" if np.signbit(v) else "nan"
    if np.isinf(v):
        return "-inf" if v < 0 else "inf"
    return "%.3f" % v


def _warning_lines(count_and_list, labels):
    n, recs = count_and_list
    out = []
    for body, kind, val in recs:
        if kind in (3, 5):
            out.append(f"[Physics] Warning: body {body + 1} produced non-finite {labels[kind]}; resetting to identity.")
        else:
            out.append(f"[Physics] Warning: body {body + 1} produced non-finite {labels[kind]} ({_c_float3(val[0])}, {_c_float3(val[1])}, {_c_float3(val[2])}); resetting to zero.")
    return out


def test_non_finite_warnings_are_the_reference_lines_in_the_reference_order(capfd, tmp_path):
    """solver.cpp:51-66 prints one line per reset while it steps; the kernels log the resets and the host prints them.  Non-finite
    velocities, positions and orientations injected into several bodies: the logged warnings, formatted like the reference's
    printf, equal the oracle's stdout line for line, through the C ABI and through the CLI (host mirror) started from a snapshot."""
    import ctypes
    import subprocess
    o, g = pair()
    for w in (o, g):
        w.scene("Pyramid")
    o.step(12); g.step(12)
    b = g.bodies().copy()
    b["linVel"][5] = (np.nan, 1.0, np.inf)
    b["angVel"][9] = (np.inf, 0.0, -np.inf)
    b["linVel"][20] = (-np.inf, -np.nan, 0.0)
    b["pos"][31] = (np.nan, 2.0, 0.0)
    b["quat"][40] = (np.nan, 0.0, 0.0, 1.0)
    b["angVel"][40] = (0.0, np.inf, 0.0)
    f = o.forces()
    o.set_bodies(b); o.set_forces(f)
    g.set_bodies(b); g.set_forces(f)
    snap = str(tmp_path / "nan.snap")
    g.w.save_snapshot(snap)
    labels = g.w.WARNING_LABELS
    ctypes.CDLL(None).fflush(None)                              # whatever earlier tests left in the C stdio buffer
    capfd.readouterr()
    total = 0
    want_all = []
    for s in range(3):
        o.step()
        ctypes.CDLL(None).fflush(None)                          # the oracle prints through C stdio
        want = [l for l in capfd.readouterr().out.splitlines() if l.startswith("[Physics] Warning")]
        g.step()
        got = _warning_lines(g.w.warnings(), labels)
        # the sign glibc prints for a NaN is its sign bit, and a NaN GENERATED by an operation is negative on x86 (the "real
        # indefinite") but positive on gfx950; the quantity is reset to zero either way, only this one character can differ
        unsigned = lambda lines: [l.replace("-nan", "nan") for l in lines]
        assert unsigned(got) == unsigned(want), f"step {s + 1}"
        assert [l for l in got if "predicted" in l or "velocity (" in l][:6] == [l for l in want if "predicted" in l or "velocity (" in l][:6]
        assert int(g.w.stats()["warnings"]) == len(want)
        total += len(want)
        want_all += want
        if s > 0:     # step 1 leaves a NaN in the (never sanitised) inertial orientation of body 40: NaN on both sides, sign as above
            H.assert_same_world(o, g, f"step {s + 1}")
        else:
            bo, bg = o.bodies(), g.bodies()
            for fld in ("pos", "quat", "linVel", "angVel"):
                assert H.same_bits(bo[fld], bg[fld]), fld
    assert total >= 6
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    out = subprocess.run([exe, "--nogfx", "--load-state", snap, "--steps", "3"], capture_output=True, text=True, check=True).stdout.splitlines()
    assert [l.replace("-nan", "nan") for l in out if l.startswith("[Physics] Warning")] == [l.replace("-nan", "nan") for l in want_all]
    # and they come before the diagnostics line of their step, as in the reference
    first_warning = next(i for i, l in enumerate(out) if l.startswith("[Physics] Warning"))
    first_diag = next(i for i, l in enumerate(out) if l.startswith("[Physics] step"))
    assert first_warning < first_diag


def test_debug_race_checker_accepts_every_schedule(monkeypatch):
    """AVBD_RACE_CHECK=1: after every topology rebuild a kernel checks that the schedule about to run is race-free —
    no two bodies of one level launch share a force; the sweep order of the data-flow schedules is a linear extension of the
    dependencies and lists every swept body once; every incidence entry belongs to its body.  It must stay silent on all
    schedules (and the results stay bit-exact)."""
    monkeypatch.setenv("AVBD_RACE_CHECK", "1")
    for env in ({}, {"AVBD_MODE": "levels"}, {"AVBD_ORDER": "levels"}, {"AVBD_GROUPS": "1"}, {"AVBD_FUSED_TOPOLOGY": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o = H.CpuWorld("orc"); o.scene("Pyramid"); o.step(25)
        g = H.HipWorld(); g.scene("Pyramid"); g.step(25)
        g.w.synchronize()                                    # a violation would surface here as AVBD_E_HIP
        H.assert_same_world(o, g, f"Pyramid under the race checker with {env}")
        for k in env:
            monkeypatch.delenv(k)
    g = H.HipWorld(); g.scene("Packed:13"); g.set_params(iterations=4); g.step(6); g.w.synchronize()
    assert int(g.w.stats()["schedule"]) == 4
    g = H.HipWorld(); g.scene("Packed:12"); g.set_params(iterations=4); g.step(6); g.w.synchronize()
    assert int(g.w.stats()["schedule"]) == 3                 # (1728 bodies: a wave each from the two-workgroups-per-CU launch)


def test_corrected_joint_sign_option_welds_hold(monkeypatch):
    """avbd_set_joint_angular_sign(ctx, 1) (no such switch in the reference): with J_angular = dC/dtheta two
    cubes welded face to face fall, land and settle as ONE rigid body, and a tilted cube welded to the world hangs still; with the
    reference's sign (default, bug-compatible — covered bit for bit by the joint fixtures) the same pair is flung apart."""
    def build(g):
        g.add_body((100, 1, 100), 0.0, 0.5, (0, -0.5, 0))
        a = g.add_body((1, 1, 1), 1.0, 0.5, (0.0, 2.0, 0.0))
        b = g.add_body((1, 1, 1), 1.0, 0.5, (1.0, 2.0, 0.0))
        c = g.add_body((1, 1, 1), 1.0, 0.5, (5.0, 3.0, 0.0), quat=(0.0, 0.0, 0.19866933, 0.98006658))
        return a, b, c
    g = H.HipWorld()
    a, b, c = build(g)
    g.w.set_joint_angular_sign(1)
    g.add_joint(a, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
    g.add_joint(-1, c, (5.0, 3.0, 0.0))
    g.step(240)
    bd = g.bodies()
    assert np.isfinite(bd["pos"]).all() and np.isfinite(bd["quat"]).all()
    rel = bd["pos"][b] - bd["pos"][a]
    assert abs(float(np.linalg.norm(rel)) - 1.0) < 2e-2, rel                      # still one cube edge apart
    assert abs(float(np.abs(np.dot(bd["quat"][a], bd["quat"][b]))) - 1.0) < 1e-3    # same orientation
    assert float(np.abs(bd["linVel"][[a, b]]).max()) < 5e-2 and 0.4 < float(bd["pos"][a][1]) < 0.6   # at rest on the ground
    assert float(np.linalg.norm(bd["pos"][c] - np.array([5.0, 3.0, 0.0], np.float32))) < 2e-2        # the world weld holds its anchor
    assert float(np.abs(bd["angVel"][c]).max()) < 5e-2
    # the default sign on the same scene: the welded pair does not stay together (the reference's behaviour)
    r = H.HipWorld()
    a, b, c = build(r)
    r.add_joint(a, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
    r.step(60)
    br = r.bodies()
    assert not (abs(float(np.linalg.norm(br["pos"][b] - br["pos"][a])) - 1.0) < 2e-2 and abs(float(np.abs(np.dot(br["quat"][a], br["quat"][b]))) - 1.0) < 1e-3)


def test_worlds_of_one_process_stepped_without_a_synchronize_in_between():
    """Two Solvers in one program (nothing in the reference forbids it): their steps are queued back to back on one device, no
    synchronize in between.  Every step's iteration loop is a launch that needs all its workgroups resident; two of them beside
    each other — or beside the other world's topology kernels — used to end in a dependency time-out in more than half of the runs.
    avbd_step / avbd_group_step calls of different contexts on one device now take turns on the device (one event per device), and
    each world ends with the bits it reaches alone."""
    scenes = ["Stress1000", "StressN:12", "Pyramid"]
    alone = []
    for s in scenes:
        w = H.HipWorld(); w.scene(s); w.step(200); w.w.synchronize()
        alone.append((w.bodies().tobytes(), w.forces().tobytes()))
        w.close()
    ws = [H.HipWorld() for _ in scenes]
    for w, s in zip(ws, scenes):
        w.scene(s)
    for k in range(40):
        for w in ws:
            w.step(5)                                    # queued, not waited for
    for w, s, ref in zip(ws, scenes, alone):
        w.w.synchronize()
        assert w.bodies().tobytes() == ref[0] and w.forces().tobytes() == ref[1], s
        w.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_a_pile_too_dense_for_the_manifold_pool_is_an_error_not_a_fault(fused):
    """548 boxes dropped into each other: 41 000 candidate pairs and more surviving manifolds than the pool of a context of this size holds.  The narrow
    phase counts the survivors it has no slot for too (that is how the overflow is seen); everything behind it must take the pool's capacity, not that
    count, as the number of manifolds — k_topology_small walked off the end of the table (found by tools/soak_small.py, SOAK_SCALE=4, seed 9003)."""
    pkg = H.load_product()
    g = H.HipWorld()
    H.random_pile(g, np.random.default_rng(9003), 548, spread=1.2, height=5.0)
    g.w.set_option("fused_front", fused); g.w.set_option("fused_topology", fused)
    with pytest.raises(pkg.AvbdError, match="pool overflow"):
        g.step(2); g.w.synchronize()


def test_pool_per_body_sizes_the_pools_for_a_dense_pile():
    """the pile of the test above with pools for 64 manifolds per body: it runs, holds more manifolds than the default pools, and equals the oracle"""
    o, g = H.CpuWorld("orc"), H.HipWorld()
    g.w.set_option("pool_per_body", 64)
    for w in (o, g):
        H.random_pile(w, np.random.default_rng(9003), 548, spread=1.2, height=5.0)
        w.set_params(iterations=5)
    o.step(1); g.step(1)
    assert int(g.w.stats()["numManifolds"]) > 16384              # (22 579 after the first step; the default pools of 549 bodies hold 16 384)
    o.step(2); g.step(2)
    H.assert_same_world(o, g, "548 boxes dropped into each other, three steps")
    # its snapshot restores into a context with default pools (the restore sizes them for what it is given) and goes on bit-identically
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "dense.snap")
        g.w.save_snapshot(path)
        h = H.HipWorld(); h.w.load_snapshot(path)
    o.step(3); h.step(3)
    H.assert_same_world(o, h, "three more steps behind a restore into default pools")
