#!/usr/bin/env python
"""Register / scratch / occupancy figures of every kernel in csrc/engine.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950; no GPU needed).
  python tools/kernel_resources.py [filter-substring ...]       prints name, VGPR, AGPR, SGPR, scratch B/lane, occupancy, LDS
tests/test_abi_cpu.py uses parse() to keep the steer kernels free of scratch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def remarks(extra=()):
    src = os.path.join(ROOT, "lqrrt_amd", "csrc", "engine.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except OSError:
        return names


def parse(text):
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append(dict(mangled=b.split("\n")[0].strip(), vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"),
                         scratch=g(r"ScratchSize \[bytes/lane\]"), occupancy=g(r"Occupancy \[waves/SIMD\]"),
                         lds=g(r"LDS Size \[bytes/block\]")))
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    return rows


if __name__ == "__main__":
    filt = sys.argv[1:]
    for r in parse(remarks()):
        if not filt or any(f in r["name"] for f in filt):
            print("%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
                r["name"][:100], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["occupancy"], r["lds"]))
