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


REMARK_FLAG = "-Rpass-analysis=kernel-resource-usage"
CACHE = os.path.join(ROOT, "build", "isa_cache")                # git-ignored; keyed by the content of every source the compile reads


def source_key(extra=()):
    """Hash of what a compile of csrc/engine.hip with `extra` depends on: csrc/*, include/*, a user header named by
    -DLQRRT_USER_SYSTEM and the flags themselves."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d in (os.path.join(ROOT, "lqrrt_amd", "csrc"), os.path.join(ROOT, "include")):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp", ".def", ".h"))]
    for x in extra:
        h.update(x.encode())
        m = re.match(r'-DLQRRT_USER_SYSTEM="?([^"]+)"?$', x)
        if m:
            files.append(m.group(1))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:24]


def cache_file(kind, extra=()):
    return os.path.join(CACHE, "%s.%s" % (source_key(extra), kind))


def prune(keep=2, protect=()):
    """The assembly files are ~70 MB each: only the newest `keep` stay (with their remarks), plus the keys in `protect`."""
    try:
        asm = sorted((f for f in os.listdir(CACHE) if f.endswith(".s")), key=lambda f: os.path.getmtime(os.path.join(CACHE, f)), reverse=True)
        for f in asm[keep:]:
            if f[:-2] in protect:
                continue
            for g in (f, f[:-2] + ".remarks"):
                if os.path.exists(os.path.join(CACHE, g)):
                    os.remove(os.path.join(CACHE, g))
    except OSError:
        pass


def store(kind, extra, text):
    os.makedirs(CACHE, exist_ok=True)
    tmp = cache_file(kind, extra) + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, cache_file(kind, extra))


def remarks(extra=()):
    """hipcc's kernel-resource-usage remarks for csrc/engine.hip (a ~3 min compile), cached by source content: __graft_entry__.build()
    captures them from the compile it runs anyway (tools/build_user_system.py), so the CPU tests that read them cost nothing after a build."""
    path = cache_file("remarks", extra)
    if os.path.exists(path):
        return open(path).read()
    src = os.path.join(ROOT, "lqrrt_amd", "csrc", "engine.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o", "/dev/null",
           REMARK_FLAG] + list(extra)
    text = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
    if "Function Name" in text:
        store("remarks", extra, text)
    return text


def device_asm_command(out, extra=()):
    src = os.path.join(ROOT, "lqrrt_amd", "csrc", "engine.hip")
    return [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only", src, "-o", out] + list(extra)


def device_asm(extra=()):
    """The gfx950 assembly of csrc/engine.hip (cached like remarks(); build() produces it beside its own compiles)."""
    path = cache_file("s", extra)
    if not os.path.exists(path):
        os.makedirs(CACHE, exist_ok=True)
        tmp = path + ".tmp%d.s" % os.getpid()
        subprocess.run(device_asm_command(tmp, extra), check=True, capture_output=True, cwd=os.path.join(ROOT, "lqrrt_amd", "csrc"))
        os.replace(tmp, path)
        prune(protect=(source_key(extra),))
    return open(path).read()


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


def private_memory_instructions(mangled_names, extra=()):
    """ISA check behind the ScratchSize remark: the steer kernels live at the SGPR limit, and the register allocator sometimes leaves
    a few dozen bytes of frame RESERVED (an emergency spill slot) that no instruction touches -- `ScratchSize 36` with nothing spilled.
    Returns {mangled name: number of scratch_* / private buffer_load|store instructions in the kernel's code}."""
    text = device_asm(extra)
    res = {}
    for name in mangled_names:
        name = name.split()[0]                                  # (the remark line carries " [-Rpass-analysis=...]" behind the symbol)
        a = text.index("\n" + name + ":")
        body = text[a:text.index("s_endpgm", a)]
        res[name] = len(re.findall(r"^\s+(scratch_(load|store)\w*|buffer_(load|store)_dword\w*\s+v\d+, (off|v\d+), s\[\d+:\d+\], (0|s\d+) (offen|offset))", body, flags=re.M))
    return res


if __name__ == "__main__":
    filt = sys.argv[1:]
    for r in parse(remarks()):
        if not filt or any(f in r["name"] for f in filt):
            print("%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
                r["name"][:100], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["occupancy"], r["lds"]))
