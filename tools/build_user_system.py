#!/usr/bin/env python
"""Builds the engine with an out-of-tree problem compiled in.

    python tools/build_user_system.py path/to/my_system.hpp -o /where/liblqrrt_mine.so

The header defines lq::UserSystem (template: examples/user_system/unicycle.hpp).  The result is a complete
liblqrrt_hip.so -- every built-in model plus LQRRT_MODEL_USER -- to be loaded instead of the stock one:
LQRRT_LIB=/where/liblqrrt_mine.so.  hipcc cross-compiles for gfx950 without a GPU (~1.5 min)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(header, out):
    header = os.path.abspath(header)
    csrc = os.path.join(ROOT, "lqrrt_amd", "csrc")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-DLQRRT_USER_SYSTEM=\"%s\"" % header, os.path.join(csrc, "engine.hip"), "-o", os.path.abspath(out)]
    subprocess.check_call(cmd, cwd=csrc)
    return os.path.abspath(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("header")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    print(build(a.header, a.out))
