"""Host mirror of the reference's object API (avbd-demo3d_amd/host): the row API on mirrors never frees them, deleting a
mirror unregisters it (tests/host/mirror_lifetime.cpp, built here against libavbd_hip.so and run with glibc's heap checks)."""
import os
import subprocess

import pytest

import harness as H

pytestmark = pytest.mark.gpu


def test_row_api_on_host_mirrors_keeps_them_alive(tmp_path):
    pkg_dir = os.path.join(H.ROOT, "avbd-demo3d_amd")
    exe = str(tmp_path / "mirror_lifetime")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-I", os.path.join(pkg_dir, "host"),
                    os.path.join(H.ROOT, "tests", "host", "mirror_lifetime.cpp"), "-L", pkg_dir, "-lavbd_hip",
                    "-Wl,-rpath," + pkg_dir, "-o", exe], check=True)
    env = dict(os.environ, MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
