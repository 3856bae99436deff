"""Corrupted snapshots and force uploads are rejected with AVBD_E_INVALID before the context is touched:
every record that can reach the device tables is validated first, and a failed load leaves the old scene stepping on."""
import os
import struct

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu

HEADER = 8 + 6 * 4 + 9 * 4        # magic, 6 ints, AvbdParams (dt, gravity[3], iterations, alpha, beta, gamma, postStabilize)


def _snapshot(tmp_path):
    a = H.HipWorld(); a.scene("Pyramid"); a.step(30)
    path = str(tmp_path / "good.snap")
    a.w.save_snapshot(path)
    return a, open(path, "rb").read()


def _expect_rejected(g, o, path):
    with pytest.raises(g.pkg.AvbdError):
        g.w.load_snapshot(path)
    o.step(2); g.step(2)                                   # the scene that was there before is intact and goes on bit for bit
    H.assert_same_world(o, g, "after a rejected snapshot")


def test_corrupted_snapshots_are_rejected_and_leave_the_context_alone(tmp_path):
    a, good = _snapshot(tmp_path)
    nb, nf = struct.unpack_from("<ii", good, 8 + 8)
    assert nb == 56 and nf > 50
    body_bytes, force_bytes = struct.unpack_from("<ii", good, 8)
    forces_at = HEADER + nb * body_bytes
    assert len(good) == forces_at + nf * force_bytes
    o = H.CpuWorld("orc"); o.scene("Stack"); o.step(5)
    g = H.HipWorld(); g.scene("Stack"); g.step(5)
    rng = np.random.default_rng(5)

    def write(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        return p

    cases = {}
    cases["truncated"] = good[:-100]
    cases["trailing bytes"] = good + b"\0" * 16
    cases["huge force count"] = good[:8 + 12] + struct.pack("<i", 2 ** 30) + good[8 + 16:]
    cases["negative body count"] = good[:8 + 8] + struct.pack("<i", -5) + good[8 + 12:]
    cases["bad magic"] = b"AVBDSNPX" + good[8:]
    # field offsets inside an AvbdForce record: kind, bodyA, bodyB, rows at 0..16; numContacts behind the seven row arrays
    k = int(rng.integers(0, nf))
    rec = forces_at + k * force_bytes
    for label, off, val in (("kind 7", 0, 7), ("bodyA out of range", 4, nb + 3), ("bodyB negative", 8, -1), ("rows 40", 12, 40),
                            ("numContacts 9", 16 + 7 * 48, 9), ("numContacts negative", 16 + 7 * 48, -2), ("bodyA huge", 4, 2 ** 30)):
        cases[label] = good[:rec + off] + struct.pack("<i", val) + good[rec + off + 4:]
    for label, data in cases.items():
        _expect_rejected(g, o, write(label.replace(" ", "_") + ".snap", data))
    # and the good file still loads
    g.w.load_snapshot(write("again.snap", good))
    a.step(3); g.step(3)
    assert a.bodies().tobytes() == g.bodies().tobytes()


def test_upload_forces_validates_its_records():
    o = H.CpuWorld("orc"); o.scene("Pyramid"); o.step(30)
    g = H.HipWorld(); g.scene("Pyramid"); g.step(30)
    f = o.forces().copy()
    for field, val in (("numContacts", 5), ("bodyB", 1000), ("bodyA", -3), ("kind", 9)):
        bad = f.copy()
        bad[field][len(bad) // 2] = val
        with pytest.raises(Exception):
            g.set_forces(bad)
        H.assert_same_world(o, g, f"after a rejected upload ({field})", check_diag=False)
    o.step(); g.step()
    H.assert_same_world(o, g, "one step after the rejected uploads")
