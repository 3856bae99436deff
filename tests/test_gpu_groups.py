"""Group schedule (csrc/avbd_groups.h): several bodies in flight per wavefront, sub-groups of 16 / 32 / 64 lanes.

`stats.schedule == 4` says the device really ran k_iterate_groups (not the launch queued behind it).  AVBD_GROUPS=1 offers
every eligible scene to it, so the small reference scenes pin it bit for bit against the oracle; by default it takes scenes
with more swept bodies than one-body-per-wave waves (> 1024)."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def _sched(g):
    return int(g.w.stats()["schedule"])


@pytest.mark.parametrize("scene,steps", [("Stack", 60), ("Pyramid", 40), ("Wall", 30), ("TwoBlockDrop", 150)])
def test_reference_scenes_in_the_group_schedule(monkeypatch, scene, steps):
    monkeypatch.setenv("AVBD_GROUPS", "1")
    o = H.CpuWorld("orc"); o.scene(scene)
    g = H.HipWorld(); g.scene(scene)
    for s in range(0, steps, 10):
        o.step(10); g.step(10)
        H.assert_same_world(o, g, f"{scene} step {s + 10} (groups)")
    assert _sched(g) == 4


def test_stress1000_steady_pile_in_the_group_schedule(monkeypatch):
    """1000 cubes in 250 waves of four sub-groups; bodies with more than 16 entries get 32-lane sub-groups."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    o = H.CpuWorld("orc"); o.scene("Stress1000"); o.step(170)
    g = H.HipWorld(); g.scene("Stress1000"); g.step(170)
    assert _sched(g) == 4
    H.assert_same_world(o, g, "Stress1000 step 170 (groups)")
    o.step(30); g.step(30)
    H.assert_same_world(o, g, "Stress1000 step 200 (groups)")
    ne = np.bincount(np.concatenate([g.forces()["bodyA"], g.forces()["bodyB"]]))
    assert ne.max() >= 1


@pytest.mark.parametrize("post,iters", [(0, 6), (1, 5), (0, 1), (1, 20)])
def test_post_stabilize_and_iteration_counts(monkeypatch, post, iters):
    monkeypatch.setenv("AVBD_GROUPS", "1")
    o = H.CpuWorld("orc"); o.scene("Packed:6"); o.set_params(postStabilize=post, iterations=iters)
    g = H.HipWorld(); g.scene("Packed:6"); g.set_params(postStabilize=post, iterations=iters)
    o.step(12); g.step(12)
    assert _sched(g) == 4
    H.assert_same_world(o, g, f"Packed:6 postStabilize={post} iterations={iters} (groups)")


def test_default_takes_scenes_beyond_one_body_per_wave():
    """Packed:13 = 2197 cubes > the 2048 waves of the two-workgroups-per-CU launch: the group schedule by default, every body resident in a sub-group.
    Packed:12 = 1728 cubes still get a wave each there (resident schedule, several-bodies form with one body per wave: 11 % faster than sixteen lanes per body)."""
    o = H.CpuWorld("orc"); o.scene("Packed:13"); o.set_params(iterations=6); o.step(14)
    g = H.HipWorld(); g.scene("Packed:13"); g.set_params(iterations=6); g.step(14)
    assert _sched(g) == 4
    H.assert_same_world_fast(o, g, "Packed:13 (groups by default)", check_rows=True)
    o = H.CpuWorld("orc"); o.scene("Packed:12"); o.set_params(iterations=6); o.step(14)
    g = H.HipWorld(); g.scene("Packed:12"); g.set_params(iterations=6); g.step(14)
    assert _sched(g) == 3
    H.assert_same_world_fast(o, g, "Packed:12 (a wave each by default)", check_rows=True)


def _plate_scene(w, k):
    """a wide dynamic plate on the ground carrying k x k unit cubes"""
    w.add_body((40, 1, 40), 0.0, 0.5, (0, -0.5, 0))
    w.add_body((8, 0.5, 8), 0.5, 0.6, (0, 0.27, 0))
    for i in range(k):
        for j in range(k):
            w.add_body((1, 1, 1), 1.0, 0.5, ((i - (k - 1) / 2) * 1.3, 1.04, (j - (k - 1) / 2) * 1.3))
    w.set_params(iterations=8)


def _plate_entries(o):
    return sum(int(r["numContacts"]) for r in o.forces() if 1 in (int(r["bodyA"]), int(r["bodyB"])))


@pytest.mark.parametrize("k,lo,hi", [(2, 17, 32), (3, 33, 64)])
def test_wide_bodies_get_wide_sub_groups(monkeypatch, k, lo, hi):
    """A plate carrying k x k cubes has more than 16 (k = 2) / 32 (k = 3) entries: a 32- / 64-lane sub-group of its own;
    the cubes keep 16 lanes."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    o = H.CpuWorld("orc"); g = H.HipWorld()
    for w in (o, g):
        _plate_scene(w, k)
    for s in range(25):
        o.step(); g.step()
        H.assert_same_world(o, g, f"plate with {k}x{k} cubes, step {s + 1} (groups)")
    assert _sched(g) == 4
    assert lo <= _plate_entries(o) <= hi, _plate_entries(o)


def test_declined_on_device_runs_the_launch_behind_it(monkeypatch):
    """More than 64 entries on one body: the device declines (GroupPlan.bad) and the queued k_iterate does the step."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    o = H.CpuWorld("orc"); g = H.HipWorld()
    for w in (o, g):
        _plate_scene(w, 5)
    for s in range(20):
        o.step(); g.step()
    assert _sched(g) != 4
    assert _plate_entries(o) > 64
    H.assert_same_world(o, g, "plate with more than 64 entries (declined)")


@pytest.mark.parametrize("scene,steps,waves", [("Pyramid", 40, 8), ("Pyramid", 40, 1), ("Wall", 30, 5), ("Stress1000", 170, 64), ("Stress1000", 160, 20)])
def test_several_bodies_per_sub_group(monkeypatch, scene, steps, waves):
    """Fewer sub-groups than bodies (AVBD_GROUPS_WAVES limits the waves the plan may use, AVBD_GROUPS_ROUNDS allows several
    bodies per sub-group): a sub-group runs whichever of its bodies is ready and parks their state between tasks; the sub-groups
    of a wave pick independently."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    monkeypatch.setenv("AVBD_GROUPS_ROUNDS", "16")
    monkeypatch.setenv("AVBD_GROUPS_WAVES", str(waves))
    o = H.CpuWorld("orc"); o.scene(scene); o.step(steps)
    g = H.HipWorld(); g.scene(scene); g.step(steps)
    assert _sched(g) == 4
    H.assert_same_world(o, g, f"{scene} step {steps} with {waves} waves (groups, several rounds)")
    o.step(5); g.step(5)
    H.assert_same_world(o, g, f"{scene} step {steps + 5} with {waves} waves (groups, several rounds)")


@pytest.mark.parametrize("post,iters", [(1, 4), (0, 1)])
def test_several_rounds_post_stabilize(monkeypatch, post, iters):
    monkeypatch.setenv("AVBD_GROUPS", "1")
    monkeypatch.setenv("AVBD_GROUPS_ROUNDS", "16")
    monkeypatch.setenv("AVBD_GROUPS_WAVES", "6")
    o = H.CpuWorld("orc"); o.scene("Packed:6"); o.set_params(postStabilize=post, iterations=iters)
    g = H.HipWorld(); g.scene("Packed:6"); g.set_params(postStabilize=post, iterations=iters)
    o.step(12); g.step(12)
    assert _sched(g) == 4
    H.assert_same_world(o, g, f"Packed:6 postStabilize={post} iterations={iters} (groups, 6 waves)")


def test_wide_bodies_with_several_rounds(monkeypatch):
    """Three classes at once and more bodies than sub-groups: 2 waves for a plate (> 32 entries) and nine cubes."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    monkeypatch.setenv("AVBD_GROUPS_ROUNDS", "16")
    monkeypatch.setenv("AVBD_GROUPS_WAVES", "2")
    o = H.CpuWorld("orc"); g = H.HipWorld()
    for w in (o, g):
        _plate_scene(w, 3)
    for s in range(25):
        o.step(); g.step()
        H.assert_same_world(o, g, f"plate with 3x3 cubes on 2 waves, step {s + 1} (groups)")
    assert _sched(g) == 4


# ---- Joint forces in the group schedule (round 5): a lane holds one Manifold contact or one half (three rows) of a Joint -----------------
def _jointed(seed, big=False):
    from test_gpu_fuzz import build_jointed
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        build_jointed(w, seed, big=big)
    return o, g


@pytest.mark.parametrize("first", [300, 320])
def test_random_jointed_scenes_in_the_group_schedule(monkeypatch, first):
    """Hard, soft and world-anchored joints + ignore markers beside the contacts, one body per sub-group: every step against the oracle."""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    for seed in range(first, first + 20):
        o, g = _jointed(seed)
        for s in range(18):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1} (groups, joints)")
        assert _sched(g) == 4 and int(g.w.stats()["numJoints"]) > 0


def test_jointed_scenes_beyond_one_body_per_wave_take_the_group_schedule(monkeypatch):
    """joints beside the contacts, by default: 1100-1900 bodies get a wave each from the two-workgroups-per-CU launch (the resident schedule carries joints), and
    beyond its 2048 waves the group schedule takes them (one Joint used to send the whole sweep to the resident schedule's parked form: 2197 cubes + 400 welds here)"""
    for seed in (9001, 9003):
        o, g = _jointed(seed, big=True)
        for s in range(10):
            o.step(); g.step()
            H.assert_same_world_fast(o, g, f"seed {seed} step {s + 1}", check_rows=True)
        assert _sched(g) == 3
    monkeypatch.setenv("AVBD_GROUPS_WAVES", "0")
    o = H.CpuWorld("orc"); o.scene("StressN:13+welds:400"); o.set_params(iterations=6)
    g = H.HipWorld(); g.scene("StressN:13+welds:400"); g.set_params(iterations=6)
    for s in range(4):
        o.step(3); g.step(3)
        H.assert_same_world_fast(o, g, f"2197 cubes + 400 welds step {3 * (s + 1)}", check_rows=True)
    assert _sched(g) == 4 and int(g.w.stats()["numJoints"]) == 400


@pytest.mark.parametrize("waves,pool,spec", [(3, 1, 0), (3, 0, 2), (3, 0, 1), (40, 1, 0), (40, 0, 0)])
def test_jointed_scenes_with_several_bodies_per_sub_group(monkeypatch, waves, pool, spec):
    """fewer sub-groups than bodies: joint halves are parked in their GForce records (owner) or come back with the dual records (the other endpoint);
    pools, a task in hand and the plain loop"""
    monkeypatch.setenv("AVBD_GROUPS", "1")
    monkeypatch.setenv("AVBD_GROUPS_ROUNDS", "16")
    monkeypatch.setenv("AVBD_GROUPS_WAVES", str(waves))
    monkeypatch.setenv("AVBD_GROUP_POOL", str(pool))
    monkeypatch.setenv("AVBD_GROUP_SPEC", str(spec))
    big = waves >= 40
    for seed in ((9001, 9003) if big else range(400, 410)):
        o, g = _jointed(seed, big=big)
        for s in range(10 if big else 14):
            o.step(); g.step()
            if big:
                H.assert_same_world_fast(o, g, f"seed {seed} step {s + 1}", check_rows=True)
            else:
                H.assert_same_world(o, g, f"seed {seed} step {s + 1} (groups, joints, {waves} waves)")
        assert _sched(g) == 4
