#!/usr/bin/env python3
"""Per-kernel resources of the shipped device code, from `make -C avbd-demo3d_amd/csrc asm` (avbd_gfx950.s).

    python tools/kernel_resources.py [avbd_gfx950.s]            table of every kernel
    python tools/kernel_resources.py avbd_gfx950.s <substr>     + instruction mix of the kernels whose name contains <substr>,
                                                                 split at the `; AVBD_MARK <name>` comments the source leaves
                                                                 (asm volatile("; AVBD_MARK x")) when built with -DAVBD_ASM_MARKS
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "").replace("avbd::", "") for o in out]


def parse(path):
    kernels = collections.OrderedDict()
    cur, body = None, []
    for line in open(path):
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and (m.group(1).startswith("_Z") or m.group(1) == "k_row_api"):
            cur = m.group(1)
            body = []
            kernels[cur] = {"body": body}
            continue
        if cur is None:
            continue
        m = re.match(r"^; (codeLenInByte|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs)\s*[:=]\s*(\d+)", line)
        if m:
            kernels[cur][m.group(1)] = int(m.group(2))
            continue
        if line.startswith("\t.amdhsa_kernel") or line.startswith("\t.section") or line.startswith("\t.set"):
            continue
        body.append(line)
    return kernels


def classify(op):
    if op.startswith("v_div_") or op.startswith("v_rcp") or op.startswith("v_sqrt") or op.startswith("v_rsq"):
        return "div/sqrt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def mix(body):
    regions = collections.OrderedDict()
    name = "(entry)"
    regions[name] = collections.Counter()
    for line in body:
        m = re.search(r"; AVBD_MARK (\S+)", line)
        if m:
            name = m.group(1)
            regions.setdefault(name, collections.Counter())
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        regions[name][classify(op)] += 1
    return regions


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "avbd-demo3d_amd", "csrc", "avbd_gfx950.s")
    want = sys.argv[2] if len(sys.argv) > 2 else None
    K = parse(path)
    names = demangle(list(K))
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s} {'code B':>8s}")
    for (k, v), n in zip(K.items(), names):
        if "codeLenInByte" not in v:
            continue
        print(f"{n[:44]:44s} {v.get('NumVgprs', 0):5d} {v.get('NumAgprs', 0):5d} {v.get('ScratchSize', 0):8d} {v.get('LDSByteSize', 0):7d} "
              f"{v.get('Occupancy', 0):4d} {v.get('codeLenInByte', 0):8d}")
    if want:
        for (k, v), n in zip(K.items(), names):
            if want not in n or "codeLenInByte" not in v:
                continue
            print(f"\n== {n} ==")
            cols = ["valu", "div/sqrt", "salu", "lds", "vmem", "waitcnt", "other"]
            print(f"{'region':28s} " + " ".join(f"{c:>8s}" for c in cols) + f" {'total':>8s}")
            for r, c in mix(v["body"]).items():
                print(f"{r[:28]:28s} " + " ".join(f"{c[x]:8d}" for x in cols) + f" {sum(c.values()):8d}")


if __name__ == "__main__":
    main()
