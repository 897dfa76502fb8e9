"""Builds the in-tree native libraries.

    libzl_amd/lib/libzlhip.so      HIP engine + C-ABI (include/zlhip.h), gfx950 only
    oracle/_build/libzl_oracle.so  CPU oracle (test infrastructure; never loaded by the product)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libzl_amd", "csrc")
LIBDIR = os.path.join(ROOT, "libzl_amd", "lib")
LIB = os.path.join(LIBDIR, "libzlhip.so")

HIP_SOURCES = ["zl_kernels.hip", "zl_engine.cpp", "zl_libzl.cpp"]
HEADERS = ["zl_types.h", "zl_plan.h", "zl_render.h", "zl_kernels.h", "zl_host.h", "zl_sched.h", "zl_handoff.h",
           os.path.join("..", "..", "include", "zlhip.h"), os.path.join("..", "..", "include", "libzl_hotpath.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libzlhip.so for gfx950)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, flags: list, verbose: bool = False) -> str:
    """A/B and diagnostic builds: libzl_amd/lib/libzlhip_<name>.so with extra -D flags (loaded through the ZLHIP_LIBRARY
    environment variable by scripts/; never by the package itself)."""
    global LIB
    saved, LIB = LIB, os.path.join(LIBDIR, f"libzlhip_{name}.so")
    try:
        return _build_engine(False, verbose, list(flags))
    finally:
        LIB = saved


def build_engine(force: bool = False, verbose: bool = False, stamps: bool = False) -> str:
    """stamps=True builds the diagnostic variant libzlhip_stamps.so (-DZL_STAMPS: per-workgroup
    timestamps in K2; used only by scripts/k2_stamps.py, never by the package)."""
    global LIB
    if stamps:
        out = os.path.join(LIBDIR, "libzlhip_stamps.so")
        saved, LIB = LIB, out
        try:
            return _build_engine(force, verbose, ["-DZL_STAMPS"])
        finally:
            LIB = saved
    return _build_engine(force, verbose, [])


# per source: the headers it includes (a change of one of them recompiles only the sources that see it)
_INC = os.path.join("..", "..", "include")
SOURCE_DEPS = {
    "zl_kernels.hip": ["zl_types.h", "zl_plan.h", "zl_render.h", "zl_kernels.h"],
    "zl_engine.cpp": ["zl_types.h", "zl_plan.h", "zl_host.h", "zl_kernels.h", os.path.join(_INC, "zlhip.h")],
    "zl_libzl.cpp": ["zl_render.h", "zl_types.h", "zl_sched.h", "zl_handoff.h", os.path.join(_INC, "zlhip.h"), os.path.join(_INC, "libzl_hotpath.h")],
}


def _write_kernel_resources(remarks: str, path: str) -> None:
    """The compiler's per-kernel resource remarks, one line per kernel: name, VGPRs, scratch bytes per lane, waves per SIMD, LDS bytes.
    tests/test_kernel_resources.py holds the kernels to what DESIGN.md states (the resident kernel without scratch memory, K2 without spills)."""
    import re
    rows, cur = [], None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0] + ("Spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
    with open(path, "w") as f:
        for r in rows:
            f.write(f"{r['name']} vgprs={r.get('VGPRs', -1)} scratch={r.get('ScratchSize', -1)} waves={r.get('Occupancy', -1)} lds={r.get('LDS', -1)} vgpr_spill={r.get('VGPRsSpill', -1)}\n")


def _build_engine(force: bool, verbose: bool, extra: list) -> str:
    """One object per source (cached under lib/obj/<library name>/), then one link: editing the host code does not recompile
    the kernels (two minutes)."""
    names = [s for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    envflags = [f for f in os.environ.get("ZL_EXTRA_HIPCC_FLAGS", "").split() if f]
    objdir = os.path.join(LIBDIR, "obj", os.path.splitext(os.path.basename(LIB))[0])
    os.makedirs(objdir, exist_ok=True)
    flagfile = os.path.join(objdir, "flags.txt")
    flags = " ".join(extra + envflags)
    alldeps = [os.path.join(CSRC, n) for n in names] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and not extra and not envflags and not _stale(LIB, alldeps):
        return LIB                      # (the GPU box receives the library without the object cache)
    if not os.path.exists(flagfile) or open(flagfile).read() != flags:
        force = True
    common = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
        # the oracle defines an un-fused rounding sequence; never contract a*b+c into fma
        "-ffp-contract=off", "-fno-fast-math",
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        "-Wall", "-Wno-unused-function",
    ] + extra + envflags
    objs, rebuilt = [], False
    for name in names:
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, os.path.splitext(name)[0] + ".o")
        deps = [src, os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in SOURCE_DEPS.get(name, HEADERS)]
        if force or _stale(obj, deps):
            cmd = common + ["-x", "hip", "-c", src, "-o", obj]
            kernels = name.endswith(".hip")
            if kernels:
                cmd.append("-Rpass-analysis=kernel-resource-usage")      # registers / scratch / occupancy of every kernel -> kernel_resources.txt
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                sys.stderr.write(res.stdout + res.stderr)
                raise RuntimeError(f"hipcc failed compiling {name}")
            if kernels:
                _write_kernel_resources(res.stderr, os.path.join(os.path.dirname(LIB), os.path.splitext(os.path.basename(LIB))[0] + "_kernel_resources.txt"))
            elif verbose and res.stderr:
                sys.stderr.write(res.stderr)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("hipcc failed linking libzlhip.so")
        with open(flagfile, "w") as f:
            f.write(flags)
    return LIB


def build_oracle(force: bool = False) -> str:
    odir = os.path.join(ROOT, "oracle")
    target = os.path.join(odir, "_build", "libzl_oracle.so")
    deps = [os.path.join(odir, "zl_oracle.c"), os.path.join(odir, "zl_oracle.h"), os.path.join(odir, "Makefile")]
    if force or _stale(target, deps):
        res = subprocess.run(["make", "-C", odir, "_build/libzl_oracle.so"], capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("building the CPU oracle failed")
    return target


def build_cpu_harness(force: bool = False) -> str:
    """Host build of the __host__ __device__ planning / per-frame code for CPU-only unit tests."""
    hdir = os.path.join(ROOT, "tests", "cpu_harness")
    target = os.path.join(hdir, "_build", "libzl_plan_host.so")
    src = os.path.join(hdir, "plan_host.cpp")
    deps = [src] + [os.path.join(CSRC, h) for h in ("zl_types.h", "zl_plan.h", "zl_render.h", "zl_host.h")] + [os.path.join(ROOT, "include", "zlhip.h")]
    if force or _stale(target, deps):
        os.makedirs(os.path.dirname(target), exist_ok=True)
        # (-Bsymbolic: the header-inline code of this library binds to ITS copies, not to libzlhip.so's when both are loaded)
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wl,-Bsymbolic",
               "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", target, src]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("building the CPU test harness failed")
    # the product's ClipCommand scheduler (zl_sched.h), host build
    t2 = os.path.join(hdir, "_build", "libzl_sched_host.so")
    src2 = os.path.join(hdir, "sched_host.cpp")
    if force or _stale(t2, [src2, os.path.join(CSRC, "zl_sched.h"), os.path.join(ROOT, "include", "zlhip.h")]):
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-Wl,-Bsymbolic",
               "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", t2, src2]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("building the CPU scheduler harness failed")
    return target


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_engine(force=force, verbose=True))
    print(build_oracle(force=force))
