"""Sharded (multi-context) path on ONE GPU: two contexts of one process driven in lock-step by avbd_group_step, which
runs exactly the schedule of the one-process-per-GPU RCCL path (exchange by device copies instead of the collective).

Two forms of that schedule give the same bits (both are Jacobi across the cut, one exchange per iteration):
  persistent  (default)  ONE launch per rank and step (k_iterate_groups<false, true>); the exchange runs beside it on a second
              stream: hipStreamWaitValue32 on the rank's iteration counter -> gather -> hipStreamWriteValue32 releases the
              tasks that read remote rows; tasks without a remote partner never wait for it.  stats.schedule == 4.
              A rank with more bodies than sub-groups keeps this form with several bodies per sub-group (k_iterate_groups<true, true>).
  per-iteration (AVBD_SHARD_PERSISTENT=0, joints / springs, more than four contexts on one device) one launch per iteration."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def two_piles(world, gap):
    world.add_body((120, 1, 40), 0.0, 0.5, (0, -0.5, 0))
    for cx in (-gap, gap):
        for y in range(3):
            for z in range(3):
                for x in range(3):
                    world.add_body((1, 1, 1), 1.0, 0.5, (cx + (x - 1) * 1.02, 0.52 + y * 1.02, (z - 1) * 1.02))
    world.set_params(iterations=12)


def sharded_pair(build):
    pkg = H.load_product()
    worlds = [H.HipWorld(), H.HipWorld()]
    for w in worlds:
        build(w)
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], 2, axis=0)
    for r, w in enumerate(worlds):
        w.w.set_ownership(2, r, owner)
    return pkg, worlds, owner


@pytest.fixture(params=["1", "0"], ids=["persistent", "per-iteration"])
def shard_mode(request, monkeypatch):
    monkeypatch.setenv("AVBD_SHARD_PERSISTENT", request.param)
    return request.param


def expect_schedule(worlds, mode):
    for w in worlds:
        sched = int(w.w.stats()["schedule"])
        assert (sched == 4) == (mode == "1"), (sched, mode)


def test_independent_shards_are_bit_exact(shard_mode):
    """Two piles that never touch each other: sharding must not change a single bit."""
    pkg, worlds, owner = sharded_pair(lambda w: two_piles(w, 20.0))
    assert set(owner[1:]) == {0, 1} and owner[0] == -1
    o = H.CpuWorld("orc"); two_piles(o, 20.0)
    for s in range(6):
        pkg.group_step([w.w for w in worlds], 5)
        o.step(5)
        bo = o.bodies()
        for r, w in enumerate(worlds):
            bg = w.bodies()
            for f in ("pos", "quat", "linVel", "angVel", "prevLinVel", "prevAngVel"):
                assert H.same_bits(bo[f], bg[f]), f"rank {r} field {f} after {5 * (s + 1)} steps"
    # every manifold lives on exactly the rank(s) that solve one of its bodies
    total = sum(int(w.w.stats()["numManifolds"]) for w in worlds)
    assert total == len(o.forces())
    expect_schedule(worlds, shard_mode)


def test_persistent_and_per_iteration_schedules_of_a_cut_pile_give_the_same_bits(monkeypatch):
    """One pile cut in two, 30 steps: the persistent launch (interior tasks run ahead of the exchange) and the one-launch-per-
    iteration schedule are the same iteration, bit for bit — on both ranks."""
    scene = "PackedGrid:4,3,8"
    finals = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AVBD_SHARD_PERSISTENT", mode)
        pkg = H.load_product()
        worlds = [H.HipWorld(), H.HipWorld()]
        for w in worlds:
            w.scene(scene)
        b = worlds[0].bodies()
        owner = pkg.partition_by_axis(b["pos"], b["invMass"], 2, axis=2)
        for r, w in enumerate(worlds):
            w.w.set_ownership(2, r, owner)
        for s in range(6):
            pkg.group_step([w.w for w in worlds], 5)
        expect_schedule(worlds, mode)
        finals[mode] = [w.bodies() for w in worlds]
        fo = [w.forces() for w in worlds]
        finals[mode + "f"] = fo
    for r in range(2):
        assert finals["1"][r].tobytes() == finals["0"][r].tobytes(), f"rank {r}: body state differs between the two schedules"
        assert finals["1f"][r].tobytes() == finals["0f"][r].tobytes(), f"rank {r}: force state differs between the two schedules"


def test_persistent_form_with_several_bodies_per_sub_group_gives_the_same_bits(monkeypatch):
    """Few resident waves (AVBD_GROUPS_WAVES): every sub-group of the persistent launch serves several bodies, out of order
    (k_iterate_groups<true, true>) — tasks with a remote partner wait for the all-gather of the previous iteration, the wave
    counts an iteration once all its bodies are past it.  Same bits as one launch per iteration, on both ranks."""
    scene = "PackedGrid:4,3,8"
    finals = {}
    for mode in ("multi", "per-iteration"):
        monkeypatch.setenv("AVBD_SHARD_PERSISTENT", "1" if mode == "multi" else "0")
        if mode == "multi":
            monkeypatch.setenv("AVBD_GROUPS_WAVES", "8")              # 8 waves per context = 32 sub-groups for 48 bodies each: two rounds
        else:
            monkeypatch.delenv("AVBD_GROUPS_WAVES", raising=False)
        pkg = H.load_product()
        worlds = [H.HipWorld(), H.HipWorld()]
        for w in worlds:
            w.scene(scene)
        b = worlds[0].bodies()
        owner = pkg.partition_by_axis(b["pos"], b["invMass"], 2, axis=2)
        for r, w in enumerate(worlds):
            w.w.set_ownership(2, r, owner)
        for s in range(6):
            pkg.group_step([w.w for w in worlds], 5)
        expect_schedule(worlds, "1" if mode == "multi" else "0")
        finals[mode] = ([w.bodies() for w in worlds], [w.forces() for w in worlds])
    for r in range(2):
        assert finals["multi"][0][r].tobytes() == finals["per-iteration"][0][r].tobytes(), f"rank {r}: body state differs"
        assert finals["multi"][1][r].tobytes() == finals["per-iteration"][1][r].tobytes(), f"rank {r}: force state differs"


def test_four_ranks_share_one_device_in_the_persistent_form():
    """Four contexts, a quarter of the device's resident waves each, all four persistent launches in flight together."""
    scene = "PackedGrid:4,4,8"
    pkg = H.load_product()
    worlds = [H.HipWorld() for _ in range(4)]
    for w in worlds:
        w.scene(scene)
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], 4, axis=2)
    for r, w in enumerate(worlds):
        w.w.set_ownership(4, r, owner)
    single = H.HipWorld(); single.scene(scene)
    pkg.group_step([w.w for w in worlds], 20)
    single.step(20)
    ref = worlds[0].bodies()
    for r in range(1, 4):
        br = worlds[r].bodies()
        assert H.same_bits(ref["pos"], br["pos"]) and H.same_bits(ref["quat"], br["quat"]) and H.same_bits(ref["linVel"], br["linVel"])
    expect_schedule(worlds, "1")
    assert np.abs(ref["pos"] - single.bodies()["pos"]).max() < 5e-2


def test_coupled_shards_stay_close_to_the_single_gpu_sweep(shard_mode):
    """One pile cut in two: Jacobi coupling across the cut (one pose exchange per iteration) — close, not identical."""
    scene = "PackedGrid:4,3,8"
    pkg = H.load_product()
    worlds = [H.HipWorld(), H.HipWorld()]
    for w in worlds:
        w.scene(scene)
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], 2, axis=2)
    for r, w in enumerate(worlds):
        w.w.set_ownership(2, r, owner)
    single = H.HipWorld(); single.scene(scene)
    pkg.group_step([w.w for w in worlds], 25)
    single.step(25)
    bs = single.bodies()
    b0, b1 = worlds[0].bodies(), worlds[1].bodies()
    assert H.same_bits(b0["pos"], b1["pos"]) and H.same_bits(b0["linVel"], b1["linVel"]), "both ranks must hold the same gathered state"
    assert np.all(np.isfinite(b0["pos"]))
    err = np.abs(b0["pos"] - bs["pos"]).max()
    assert err < 5e-2, f"max position deviation from the single-GPU sweep after 25 steps: {err}"
    cut = sum(int(w.w.stats()["numManifolds"]) for w in worlds)
    assert cut >= int(single.w.stats()["numManifolds"])  # cut manifolds are replicated on both sides
    expect_schedule(worlds, shard_mode)


def test_rccl_single_rank_schedule_is_bit_exact(monkeypatch, shard_mode):
    """The RCCL code path (communicator, all-gather every iteration — on the second stream beside the persistent launch, or
    between the per-iteration launches) with one rank: same bits as the fused launch."""
    monkeypatch.setenv("AVBD_FORCE_SHARDED", "1")
    pkg = H.load_product()
    g = H.HipWorld(); g.scene("Pyramid")
    monkeypatch.delenv("AVBD_FORCE_SHARDED")
    b = g.bodies()
    g.w.set_ownership(1, 0, np.where(b["invMass"] > 0, 0, -1).astype(np.int32))
    g.w.comm_init(1, 0, pkg.comm_unique_id())
    o = H.CpuWorld("orc"); o.scene("Pyramid")
    for s in range(4):
        g.step(10); o.step(10)
        H.assert_same_world(o, g, f"rccl 1-rank after {10 * (s + 1)} steps")
    expect_schedule([g], shard_mode)


def test_profiling_the_persistent_form_of_a_rank_with_a_communicator(monkeypatch):
    """Round 5: with profiling on, the persistent sharded launch used to carry start / stop events (and the tail's span an event behind it); in about half of all
    runs the kernel then crawled or stopped — a dependency time-out after 4 s — while the exchange stream's waits were in flight (tools/force_dist_flake.py; it made
    bench.py's sharded forms fail in their profiled leg).  The launch now stamps its own span with the device's clock and no event is recorded behind it: 40
    profiled steps of a landing pile go through, the span is plausible, and the bits are the unprofiled run's."""
    monkeypatch.setenv("AVBD_FORCE_SHARDED", "1")
    monkeypatch.setenv("AVBD_SHARD_PERSISTENT", "1")
    pkg = H.load_product()
    worlds = []
    for profiled in (True, False):
        g = H.HipWorld(); g.scene("StressN:12")
        b = g.bodies()
        g.w.set_ownership(1, 0, np.where(b["invMass"] > 0, 0, -1).astype(np.int32))
        g.w.comm_init(1, 0, pkg.comm_unique_id())
        g.step(300)
        if profiled:
            g.w.profile_enable(True); g.w.profile_reset()
        g.step(40); g.w.synchronize()
        if profiled:
            p = g.w.profile(); g.w.profile_enable(False)
            assert int(p["launches_primal"]) == 40 and int(p["steps"]) == 40
            assert 0.2 < float(p["ms_primal"]) / 40 < 5.0, float(p["ms_primal"]) / 40      # ~0.7 ms per launch; a crawl was 135 ms
        worlds.append(g)
    assert int(worlds[0].w.stats()["schedule"]) == 4
    assert H.same_bits(worlds[0].bodies(), worlds[1].bodies())


def _two_big_piles(w, n=13, gap=25.0):
    w.add_body((100, 1, 100), 0.0, 0.5, (0, -0.5, 0))
    for cx in (-gap, gap):
        for y in range(n):
            for z in range(n):
                for x in range(n):
                    w.add_body((1, 1, 1), 1.0, 0.5, (cx + (x - n / 2) * 1.02, 0.52 + y * 1.02, (z - n / 2) * 1.02))
    w.set_params(iterations=10)


def test_independent_shards_of_a_large_scene_are_bit_exact(shard_mode):
    """4 394 cubes in two piles that never touch, one pile per rank: beyond 4096 bodies every rank runs the grid broadphase and
    the multi-launch topology, and its 2 197 swept bodies take the resident schedule with several bodies per wave, one launch per
    iteration — the combination an 8-GPU run of a large scene uses.  Independent shards must not change a single bit."""
    pkg, worlds, owner = sharded_pair(_two_big_piles)
    assert len(owner) == 4395 and set(owner[1:]) == {0, 1}
    o = H.CpuWorld("orc"); _two_big_piles(o)
    for s in range(3):
        pkg.group_step([w.w for w in worlds], 1)
        o.step(1)
        bo = o.bodies()
        for r, w in enumerate(worlds):
            bg = w.bodies()
            for f in ("pos", "quat", "linVel", "angVel"):
                assert H.same_bits(bo[f], bg[f]), f"rank {r} field {f} after step {s + 1}"
    assert all(int(w.w.stats()["schedule"]) == (4 if shard_mode == "1" else 3) for w in worlds)
    assert sum(int(w.w.stats()["numManifolds"]) for w in worlds) == len(o.forces())


def _cut_pile_finals(monkeypatch, env, steps=6, substeps=5, exchange_floats=None, mismatches=None):
    """two ranks on a pile cut along z, avbd_group_step; returns per-rank (bodies, forces) bytes"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pkg = H.load_product()
    worlds = [H.HipWorld(), H.HipWorld()]
    for w in worlds:
        w.scene("PackedGrid:4,3,8")
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], 2, axis=2)
    for r, w in enumerate(worlds):
        w.w.set_ownership(2, r, owner)
        if exchange_floats is not None:
            w.w.set_exchange_floats(exchange_floats)
    for s in range(steps):
        pkg.group_step([w.w for w in worlds], substeps)
    out = [(w.bodies().tobytes(), w.forces().tobytes()) for w in worlds]
    sched = [int(w.w.stats()["schedule"]) for w in worlds]
    if mismatches is not None:
        mismatches.extend(w.w.exchange_mismatches() for w in worlds)
    for w in worlds:
        w.close()
    for k in env:
        monkeypatch.delenv(k)
    return out, sched


def test_exchange_stream_gated_by_kernels_gives_the_same_bits(monkeypatch):
    """AVBD_SHARD_GATE=kernel (a form bench.py --gpus N measures): the exchange stream waits for an iteration and publishes the
    gather word with one-wave kernels instead of hipStreamWaitValue32 / hipStreamWriteValue32.  Same iteration, same bits, on both
    ranks, as the stream-value form and as one launch per iteration."""
    a, sa = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": "1"})
    b, sb = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": "1", "AVBD_SHARD_GATE": "kernel"})
    c, sc = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": "0"})
    assert sa == sb == [4, 4], (sa, sb)
    for r in range(2):
        assert a[r] == b[r], f"rank {r}: kernel-gated exchange differs from the stream-value form"
        assert a[r] == c[r], f"rank {r}: persistent form differs from one launch per iteration"


def test_stamp_wrap_in_the_persistent_sharded_form(monkeypatch):
    """The stamps of the exchange records wrap every 2^24 steps; the persistent sharded form also compares them with the two words
    of its exchange stream (iterations done / all-gathers landed), which must start from zero again with the records (ADVICE round
    2).  AVBD_RESIDENT_SERIAL starts the contexts twelve steps before the wrap: 30 steps across it give the bits of a run that
    never wraps, on both ranks."""
    a, sa = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": "1"})
    b, sb = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": "1", "AVBD_RESIDENT_SERIAL": str((1 << 24) - 12)})
    assert sa == sb == [4, 4]
    for r in range(2):
        assert a[r] == b[r], f"rank {r}: the run across the stamp wrap differs"


@pytest.mark.parametrize("form", ["1", "0"], ids=["persistent", "per-iteration"])
def test_thirteen_float_exchange_gathers_the_velocities_every_rank_re_derived(monkeypatch, form):
    """avbd_set_exchange_floats(13) (pos3 + quat4 + linVel3 + angVel3): besides the poses of every iteration the ranks
    gather the velocities once per step.  Every rank has re-derived them from the gathered poses already (solver.cpp:434-469), so
    the rows that arrive equal what they overwrite bit for bit — the kernel that stores them counts the words that differ — and the
    run ends with the bits of the 8-float exchange.  48 bodies per rank x 30 steps = 1 440 rows compared on each rank."""
    mm = []
    a, sa = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": form})
    b, sb = _cut_pile_finals(monkeypatch, {"AVBD_SHARD_PERSISTENT": form}, exchange_floats=13, mismatches=mm)
    assert sa == sb == ([4, 4] if form == "1" else [3, 3]), (sa, sb)
    assert [m[0] for m in mm] == [0, 0], mm
    assert all(m[1] == 48 * 30 for m in mm), mm
    for r in range(2):
        assert a[r] == b[r], f"rank {r}: the 13-float exchange changed the result"
    pkg = H.load_product()
    w = H.HipWorld()
    with pytest.raises(Exception):
        w.w.set_exchange_floats(7)
    w.close()
