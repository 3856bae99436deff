"""avbd_set_option (include/avbd_hip.h, ABI 4): every key that selects a schedule or a form of a kernel, set through the API — not the
environment — and the result compared with the oracle bit for bit: options choose HOW the reference's sweep (solver.cpp:340-431) is run,
never what it computes.  VERDICT round 3 listed the knobs no test touched (place, group_spec, group_spec_rounds, single_kernel,
resident_patience); all of them are in the tables below."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def _run(scene, steps, options, params=None, check=H.assert_same_world):
    o = H.CpuWorld("orc"); o.scene(scene)
    g = H.HipWorld(); g.scene(scene)
    if params:
        o.set_params(**params); g.set_params(**params)
    for k, v in options.items():
        g.w.set_option(k, v)
        assert g.w.get_option(k) == v, k
    for s in range(0, steps, 10):
        n = min(10, steps - s)
        o.step(n); g.step(n)
        check(o, g, f"{scene} step {s + n} with {options}")
    return g


def test_keys_are_enumerable_and_unknown_keys_fail():
    pkg = H.load_product()
    keys = pkg.option_keys()
    assert len(keys) == len(set(keys)) >= 25
    for k in ("place", "place_every", "group_spec", "group_spec_rounds", "single_kernel", "resident_patience", "group_dep", "groups", "mode", "shard_exact"):
        assert k in keys
    w = pkg.World(0)
    with pytest.raises(pkg.AvbdError):
        w.set_option("no_such_option", 1)
    for k in keys:                                          # every key can be read back
        w.get_option(k)
    w.close()


# one body per wave (Stress1000's schedule): k_iterate_single and what replaces it
@pytest.mark.parametrize("options,schedule", [
    ({"place": 0}, 2), ({"place": 1}, 2), ({"single_kernel": 0}, 2), ({"single_kernel": 0, "place": 0}, 2),
    ({"resident_quiet": 0}, 2), ({"resident_quiet": 900}, 2), ({"resident": 0}, 1), ({"iter_mode": 2}, 1),
    ({"order_by_index": 0}, 2), ({"fused_topology": 0}, 2), ({"fused_front": 0}, 2), ({"fused_front": 0, "fused_topology": 0, "grid_broadphase": 0}, 2),
    ({"iter_blocks": 64, "resident_multi": 1}, None), ({"race_check": 1}, 2),
    ({"place_every": 1}, 2), ({"place_every": 3}, 2), ({"place_every": 1000}, 2),
    ({"front_wide_from": 0}, 2), ({"front_wide_from": 1 << 20}, 2), ({"grid_from": 2}, 2), ({"grid_from": 2, "fused_topology": 0}, 2),
])
def test_one_body_per_wave_options(options, schedule):
    g = _run("Pyramid", 40, options)
    if schedule is not None:
        assert int(g.w.stats()["schedule"]) == schedule, (options, int(g.w.stats()["schedule"]))


def test_a_kept_placement_is_rebuilt_when_the_swept_bodies_change():
    """place_every: the wave placement of the one-body-per-wave launch is kept for several steps while the same bodies are swept; a body added
    between two steps (one more wave, another sweep order) must rebuild it at once — a stale table would leave the new body without a wave"""
    o = H.CpuWorld("orc"); o.scene("Pyramid")
    g = H.HipWorld(); g.scene("Pyramid"); g.w.set_option("place_every", 50)
    o.step(12); g.step(12)
    for w in (o, g):
        w.add_body((1, 1, 1), 1.0, 0.5, (0.3, 9.0, 0.2))
    o.step(9); g.step(9)
    H.assert_same_world(o, g, "a body added at step 12 under a placement kept for 50 steps")
    for w in (o, g):
        w.add_body((0.5, 0.5, 0.5), 0.0, 0.5, (3.0, 0.25, 3.0))          # a static one: the swept bodies stay the same
        w.add_body((0.7, 0.7, 0.7), 2.0, 0.5, (0.0, 11.0, 0.0))
    o.step(14); g.step(14)
    H.assert_same_world(o, g, "two more at step 21")
    assert int(g.w.stats()["schedule"]) == 2


def test_mode_levels_is_chosen_before_the_first_step():
    g = _run("Stack", 30, {"mode": 0})
    assert int(g.w.stats()["schedule"]) == 0
    pkg = H.load_product()
    with pytest.raises(pkg.AvbdError):
        g.w.set_option("mode", 1)


# several bodies per wave of the resident schedule (more swept bodies than waves: iter_blocks makes the device small)
@pytest.mark.parametrize("options", [
    {"iter_blocks": 8, "groups": 0, "resident_ooo": 0}, {"iter_blocks": 8, "groups": 0, "resident_ooo": 1, "resident_patience": 0},
    {"iter_blocks": 8, "groups": 0, "resident_ooo": 1, "resident_patience": 2000}, {"iter_blocks": 8, "groups": 0, "resident_force_lite": 1},
    {"iter_blocks": 8, "groups": 0, "resident_multi": 0},
])
def test_several_bodies_per_wave_options(options):
    g = _run("Packed:6", 20, options, params=dict(iterations=8))
    assert int(g.w.stats()["schedule"]) in (1, 3)


# the group schedule: one body per sub-group, several bodies per sub-group (groups_waves makes the device small), the task in hand,
# the arrival counters
@pytest.mark.parametrize("options", [
    {"groups": 1}, {"groups": 1, "groups_waves": 16}, {"groups": 1, "groups_waves": 16, "group_dep": 0},
    {"groups": 1, "groups_waves": 16, "group_dep": 1, "group_dep_keep": 0}, {"groups": 1, "groups_waves": 16, "group_spec": 1},
    {"groups": 1, "groups_waves": 16, "group_spec": 2}, {"groups": 1, "groups_waves": 16, "group_spec": 0, "group_spec_rounds": 0},
    {"groups": 1, "groups_waves": 16, "group_spec": 1, "group_spec_rounds": 16, "group_dep": 0}, {"groups": 1, "groups_waves": 8, "group_spec_rounds": 16},
    {"groups": 1, "groups_waves": 8, "group_dep": 0, "group_spec": 2}, {"groups": 1, "groups_waves": 32, "groups_rounds": 4},
])
def test_group_schedule_options(options):
    g = _run("Packed:6", 20, options, params=dict(iterations=8))
    assert int(g.w.stats()["schedule"]) == 4, (options, int(g.w.stats()["schedule"]))


def test_group_schedule_options_on_a_falling_pile():
    """512 falling cubes through their landing with 8 waves x 4 sub-groups x 16 bodies: counters and task-in-hand forms, contacts that come and go"""
    for options in ({"groups": 1, "groups_waves": 8}, {"groups": 1, "groups_waves": 8, "group_dep": 0}, {"groups": 1, "groups_waves": 8, "group_spec": 1, "group_spec_rounds": 16}):
        o = H.CpuWorld("orc"); o.scene("StressN:8"); o.step(125)
        g = H.HipWorld(); g.scene("StressN:8")
        for k, v in options.items():
            g.w.set_option(k, v)
        g.step(125)
        for s in range(3):
            o.step(15); g.step(15)
            H.assert_same_world_fast(o, g, f"StressN:8 step {140 + 15 * s} with {options}", check_rows=True)
        assert int(g.w.stats()["schedule"]) == 4 and int(g.w.stats()["numManifolds"]) > 200


@pytest.mark.parametrize("lanes,order", [(32, 1), (64, 1), (64, 0)])
def test_narrow_phase_lanes_on_a_grid_scene(lanes, order):
    """k_narrowphase<32> / <64> (scenes beyond 4096 bodies; 64 lanes by default from 10 000 bodies on): 4 913 falling cubes into their landing, both forms
    against the oracle — manifolds, contacts and warm-started rows bit for bit (the 64-lane form keeps last step's lambda / penalty in registers and clips
    into eight-vertex buffers).  narrow_order: the kernel walks last step's manifolds class by class (3-4 contact points / 2 / 1 on a face / edge contact)
    so that a wave's lanes run one path — only the order in which lanes pick manifolds changes."""
    o = H.CpuWorld("orc"); o.scene("StressN:17"); o.step(128)
    g = H.HipWorld(); g.scene("StressN:17")
    g.w.set_option("narrow_lanes", lanes)
    g.w.set_option("narrow_order", order)                  # last step's manifolds class by class (default) or slot by slot
    g.step(128)
    for s in range(4):
        o.step(6); g.step(6)
        H.assert_same_world_fast(o, g, f"StressN:17 step {134 + 6 * s} with {lanes} lanes, narrow_order={order}", check_rows=True)
    assert int(g.w.stats()["numManifolds"]) > 1500
