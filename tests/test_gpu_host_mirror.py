"""Host mirror of the reference's object API (avbd-demo3d_amd/host): the row API on mirrors never frees them, deleting a
mirror unregisters it (tests/host/mirror_lifetime.cpp, built here against libavbd_hip.so and run with glibc's heap checks)."""
import os
import subprocess

import pytest

import harness as H

pytestmark = pytest.mark.gpu


def _build_and_run(tmp_path, name, args=()):
    pkg_dir = os.path.join(H.ROOT, "avbd-demo3d_amd")
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-I", os.path.join(pkg_dir, "host"),
                    os.path.join(H.ROOT, "tests", "host", name + ".cpp"), "-L", pkg_dir, "-lavbd_hip",
                    "-Wl,-rpath," + pkg_dir, "-o", exe], check=True)
    env = dict(os.environ, MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    r = subprocess.run([exe, *args], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_row_api_on_host_mirrors_keeps_them_alive(tmp_path):
    _build_and_run(tmp_path, "mirror_lifetime")


def test_user_defined_force_subclasses_run_through_host_evaluated_rows(tmp_path):
    """solver.h:102-108: a Force subclass written by the user (getRowCount / initialize / computeConstraint / computeDerivatives)
    keeps its virtuals on the host; the device fetches its rows before every visit.  A user re-statement of the distance
    constraint must give the same bits as the built-in Spring, and the runs of Tether and of the five-row Pin (with a Joint and an
    IgnoreCollision beside it) must give the bits the REFERENCE computes for the same classes: tests/host/user_forces.h compiles
    against /root/reference/source as well (oracle/user_force_ref.cpp -> tests/golden/user_force_states.bin, make_golden.py).
    The last pinned run holds two `Fuse`s whose initialize() returns false at steps 12 and 25: the solver deletes them
    (solver.cpp:274-279; destructor calls and the length of the force list are part of the pinned floats), the pair starts colliding."""
    fixture = os.path.join(H.ROOT, "tests", "golden", "user_force_states.bin")
    assert os.path.exists(fixture), ("tests/golden/user_force_states.bin is missing: it is a tracked fixture (see .gitignore's exception for "
                                     "tests/golden/*.bin); regenerate it with `make -C oracle ref && python tests/golden/make_golden.py`")
    import hashlib, json
    meta = json.load(open(os.path.join(H.ROOT, "tests", "golden", "golden_meta.json")))
    assert hashlib.md5(open(fixture, "rb").read()).hexdigest() == meta["user_force_states.bin"]["md5"]
    _build_and_run(tmp_path, "user_force", [fixture])
