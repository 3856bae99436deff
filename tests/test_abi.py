"""The C-ABI library loads without a GPU, exports every symbol include/avbd_hip.h declares, its structs have the
sizes the Python binding assumes, and it FAILS LOUDLY (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import harness as H

pkg = H.load_product()
HEADER = os.path.join(H.ROOT, "include", "avbd_hip.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    return pkg.lib()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avbd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/avbd_hip.h but not exported: {missing}"
    assert sorted(set(pkg.EXPORTS)) == names, "the Python binding's EXPORTS list and the header disagree"


def test_abi_version_and_defaults(lib):
    assert lib.avbd_abi_version() == 5
    p = np.zeros(1, pkg.PARAMS_DT)
    assert lib.avbd_default_params(p.ctypes.data_as(C.c_void_p)) == 0
    # Solver::defaultParams, solver.cpp:240-248
    assert p[0]["iterations"] == 10 and p[0]["postStabilize"] == 0
    assert p[0]["dt"] == np.float32(1.0) / np.float32(60.0)
    assert list(p[0]["gravity"]) == [0.0, -10.0, 0.0]
    assert p[0]["alpha"] == np.float32(0.95) and p[0]["beta"] == np.float32(100000.0) and p[0]["gamma"] == np.float32(0.99)


def test_struct_sizes_match_the_header():
    src = '#include <stdio.h>\n#include "avbd_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(AvbdParams), sizeof(AvbdDiagnostics),' \
          ' sizeof(AvbdBody), sizeof(AvbdContact), sizeof(AvbdForce), sizeof(AvbdStepStats), sizeof(AvbdProfile), sizeof(void*)); return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(H.ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        sizes = [int(x) for x in subprocess.run([os.path.join(d, "t")], capture_output=True, check=True, text=True).stdout.split()]
    want = [pkg.PARAMS_DT.itemsize, pkg.DIAG_DT.itemsize, pkg.BODY_DT.itemsize, pkg.CONTACT_DT.itemsize, pkg.FORCE_DT.itemsize,
            pkg.STATS_DT.itemsize, pkg.PROFILE_DT.itemsize, 8]
    assert sizes == want
    # the harness records (oracle/harness_api.h) and the product records are the same wire layout
    assert H.BODY_DT.itemsize == pkg.BODY_DT.itemsize and H.FORCE_DT.itemsize == pkg.FORCE_DT.itemsize


def test_no_cpu_fallback_without_a_device(lib):
    """In a container without /dev/kfd the product must refuse to run rather than fall back to any CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.avbd_create(0, C.byref(h))
    assert rc in (-4, -2) and not h.value
    with pytest.raises(pkg.AvbdError):
        pkg.World(0)


def test_product_never_touches_the_oracle():
    """Nothing under avbd-demo3d_amd/ or include/ may include, link or import anything under oracle/."""
    bad = []
    for root in (os.path.join(H.ROOT, "avbd-demo3d_amd"), os.path.join(H.ROOT, "include")):
        for dp, _, fns in os.walk(root):
            for fn in fns:
                if fn.endswith((".h", ".hip", ".cpp", ".py", "Makefile")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"oracle/|liboracle|harness_api|avbd_oracle", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out and "avbd_ref" not in out


def test_bench_gpus_flag_spawns_one_rank_per_gpu_before_touching_the_gpu():
    """`bench.py --gpus N` without torchrun's environment starts N child ranks (python -m torch.distributed.run ... bench.py) and
    relays them; with an environment that contradicts --gpus it refuses instead of silently running one GPU."""
    import json
    import sys
    bench = os.path.join(H.ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "7", "--warmup", "3"], capture_output=True, text=True,
                         env=dict(env, AVBD_BENCH_SPAWN_ECHO="1"), check=True).stdout
    cmd = json.loads(out.strip().splitlines()[-1])["spawn"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"] and cmd[-7].endswith("bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_option_keys_are_enumerable_without_a_device(lib):
    """avbd_option_count / avbd_option_key need no GPU; every key is documented in DESIGN.md's option table (section 7b) or in the README"""
    lib.avbd_option_key.restype = C.c_char_p
    lib.avbd_option_key.argtypes = [C.c_int]
    n = lib.avbd_option_count()
    keys = [lib.avbd_option_key(i).decode() for i in range(n)]
    assert n >= 30 and len(set(keys)) == n and lib.avbd_option_key(n) is None and lib.avbd_option_key(-1) is None
    text = open(os.path.join(H.ROOT, "DESIGN.md")).read()
    missing = [k for k in keys if ("`" + k + "`") not in text]
    assert not missing, f"options without an entry in DESIGN.md's option table: {missing}"
    # without a context the setters fail cleanly
    assert lib.avbd_set_option(None, b"place", 1) < 0


def test_committed_bench_line_and_profiles_keep_the_contract():
    """profiles/r6_bench_default.json is a line bench.py printed on an MI355X: the keys the driver reads, the two objects this tier adds (roofline with
    achieved / peak / frac / traffic, cpu_baseline with kind / cores / sample), the nested driver-timed lines of the larger scenes and of the fast sweep
    (each with value, roofline and a measured tolerance); the rocprofv3 summaries it cites exist, name the same dominant kernels and record the id of
    the library sources they were collected on — the id of the sources in this tree (tools/source_id.py)."""
    import json
    import subprocess
    import sys
    prof = os.path.join(H.ROOT, "profiles")
    line = json.load(open(os.path.join(prof, "r6_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "steps/s" and line["n_gpus"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None and "workload" in line["config"]
    assert abs(line["value"] * line["ms_per_step"] - 1000.0) < 1.0
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["sample"] and c.get("parity") == "bit-exact"
    for nested in ("strong_scaling_n1", "roofline_8k", "joints_n1"):       # (joints_n1: BASELINE configs[4]'s scene on one GPU, round 5)
        n = line["config"][nested]
        assert n["ms_per_step"] > 0 and abs(n["roofline"]["frac"] - n["roofline"]["achieved"] / n["roofline"]["peak"]) < 1e-4 and n["roofline"]["traffic"] > 0
    assert line["config"]["joints_n1"]["joints"] == 10000 and line["config"]["joints_n1"]["schedule"] == 4
    # round 6: the fast sweep beside the exact mode — never the headline value, each with its own roofline and the tolerance measured in that run
    assert line["config"]["mode"].startswith("exact")
    for nested in ("fast_1000", "fast_8k", "fast_64k"):
        n = line["config"][nested]
        assert n["schedule"] == 5 and n["mode"].startswith("fast") and abs(n["value"] * n["ms_per_step"] - 1000.0) < 1.0 and 1 <= n["colours"] <= 64
        assert n["roofline"]["kernel"] == "k_fast_sweep" and abs(n["roofline"]["frac"] - n["roofline"]["achieved"] / n["roofline"]["peak"]) < 1e-4 and n["roofline"]["traffic"] > 0
        t = n["tolerance"]
        assert t["within_bound"] is True and t["measured"]["dpos_m"] <= t["bound"]["dpos_m"] and t["measured"]["drot_rad"] <= t["bound"]["drot_rad"]
    assert line["config"]["fast_1000"]["value"] > line["value"] and line["config"]["fast_64k"]["ms_per_step"] < line["config"]["strong_scaling_n1"]["ms_per_step"]
    w = line["config"]["fast_64k_welds"]
    assert w["joints"] == 10000 and w["schedule"] == 5 and w["tolerance"]["bound"] is None
    sid = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "source_id.py")], capture_output=True, text=True, check=True).stdout.strip()
    for tag, kernel in (("stress1000", "k_iterate_single"), ("8k", "k_iterate_groups<false, 0, false>"), ("64k", "k_iterate_groups<true, 0, false>"), ("64k_welds", "k_iterate_groups<true, 0, true>"),
                        ("fast_1000", "k_fast_sweep<1, false>"), ("fast_8k", "k_fast_sweep<1, false>"), ("fast_64k", "k_fast_sweep<0, false>")):
        s = json.load(open(os.path.join(prof, f"r6_{tag}_cli_summary.json")))
        assert kernel in s["dominant_kernel"], (tag, s["dominant_kernel"])
        assert {"FETCH_SIZE", "WRITE_SIZE"} <= set(s["counters_per_dispatch_of_dominant_kernel"])
        assert os.path.exists(os.path.join(prof, f"r6_{tag}_kernel_stats.csv"))
        assert s["source_id"] == sid, f"profiles/r6_{tag}_cli_summary.json was collected on other sources ({s['source_id']} != {sid}): re-run profiles/run_all_r6.sh"
    # the resource table and the instruction budget of the round are those of the same sources
    for name in ("r6_kernel_resources.txt", "r6_instruction_budget.txt"):
        head = open(os.path.join(prof, name)).readline()
        assert f"source_id {sid}" in head, f"profiles/{name} was made from other sources: {head.strip()[-40:]}"
