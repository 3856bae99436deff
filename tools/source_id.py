"""sha256 over the sources libavbd_hip.so is built from (csrc/*.h, *.hip, *.inc, the Makefile, the host mirror, the C header): what a committed
rocprofv3 summary under profiles/ records (profiles/collect_cli.sh) and bench.py / tests/test_abi.py compare with the tree they run in."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    pk = os.path.join(ROOT, "avbd-demo3d_amd")
    files = []
    for pat in ("csrc/*.h", "csrc/*.hip", "csrc/*.inc", "csrc/Makefile", "host/*.h", "host/*.cpp"):
        files += glob.glob(os.path.join(pk, pat))
    files.append(os.path.join(ROOT, "include", "avbd_hip.h"))
    return sorted(f for f in files if "_tmp" not in os.path.basename(f))


def source_id():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode()); h.update(b"\0")
        h.update(open(f, "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_id())
