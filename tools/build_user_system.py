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
    define = "-DLQRRT_USER_SYSTEM=\"%s\"" % header
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           define, os.path.join(csrc, "engine.hip"), "-o", os.path.abspath(out)]
    # By-products for the CPU tests that pin registers / occupancy / ISA (tests/test_abi_cpu.py via tools/kernel_resources.py): the
    # resource-usage remarks of THIS compile, and the device assembly from a compile that runs beside it -- so that those tests
    # read a cache instead of compiling engine.hip two more times (~3 min each).
    import kernel_resources as kr
    asm = None
    try:
        if not os.path.exists(kr.cache_file("s", [define])):
            os.makedirs(kr.CACHE, exist_ok=True)
            asm_tmp = kr.cache_file("s", [define]) + ".tmp%d.s" % os.getpid()
            asm = (subprocess.Popen(kr.device_asm_command(asm_tmp, [define]), cwd=csrc, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), asm_tmp)
    except OSError:
        asm = None
    done = subprocess.run(cmd + [kr.REMARK_FLAG], cwd=csrc, stderr=subprocess.PIPE, text=True)
    if done.returncode != 0:
        sys.stderr.write(done.stderr)
        if asm is not None:
            asm[0].wait()
            if os.path.exists(asm[1]):
                os.remove(asm[1])
        raise subprocess.CalledProcessError(done.returncode, cmd)
    # the compiler's own diagnostics (warnings in the user's header: the one place users write device code) are passed on; the
    # resource-usage remarks asked for above are not
    noise = ("remark:", "-Rpass-analysis", "argument unused during compilation", "In file included from")
    keep, skip = [], 0
    for ln in done.stderr.splitlines():
        if any(t in ln for t in noise):
            skip = 2 if "remark:" in ln else 0                      # a remark is followed by its source line and a caret line
            continue
        if skip > 0 and (ln.strip().startswith(("|", "^")) or "|" in ln[:12]):
            skip -= 1
            continue
        skip = 0
        keep.append(ln)
    if any(ln.strip() for ln in keep):
        sys.stderr.write("\n".join(keep) + "\n")
    if "Function Name" in done.stderr:
        kr.store("remarks", [define], done.stderr)
    if asm is not None:
        if asm[0].wait() == 0:
            os.replace(asm[1], kr.cache_file("s", [define]))
            kr.prune(protect=(kr.source_key([define]),))
        elif os.path.exists(asm[1]):
            os.remove(asm[1])
    return os.path.abspath(out)


def build_oracle(header, out):
    """The same header compiled for the host (g++, oracle/user_model_shim.cpp): the callbacks the sequential C oracle calls for
    LQRRT_MODEL_USER (oracle/coracle.use_user_model), so that an out-of-tree problem is checked bit for bit like the built-in ones."""
    header = os.path.abspath(header)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared",
           "-DLQRRT_USER_SYSTEM=\"%s\"" % header, os.path.join(ROOT, "oracle", "user_model_shim.cpp"), "-o", os.path.abspath(out), "-lm"]
    subprocess.check_call(cmd)
    return os.path.abspath(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("header")
    ap.add_argument("-o", "--out", default=None, help="the engine with the problem compiled in (hipcc, ~1.5 min)")
    ap.add_argument("--oracle", default=None, help="also / only: the host build of the same header for the sequential C oracle")
    a = ap.parse_args()
    if not a.out and not a.oracle:
        ap.error("give -o and / or --oracle")
    if a.out:
        print(build(a.header, a.out))
    if a.oracle:
        print(build_oracle(a.header, a.oracle))
