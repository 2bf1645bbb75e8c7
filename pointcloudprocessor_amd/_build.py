"""Build recipe of libpcp_hip.so (hipcc, gfx950 only, in-tree).

The shared library is plain HIP runtime + C ABI: no torch types, no Python.
`build()` is what __graft_entry__.build() calls; it cross-compiles without a
GPU.  -ffp-contract=off is part of the numerical contract (pcp_device.hpp).
"""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpcp_hip.so")
INCLUDE = os.path.join(_ROOT, "include")

SOURCES = ["pcp_context.hip", "pcp_colour.hip", "pcp_mls.hip", "pcp_nid.hip", "pcp_hpr.hip"]
HEADERS = ["pcp_internal.hpp", "pcp_device.hpp", "pcp_scan.hpp", "pcp_hsv.hpp", "pcp_exact.hpp"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wextra",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_sha256() -> str:
    """SHA-256 over everything the library is made from (sources, headers, the C header, the compiler flags): two libraries
    with the same value hold the same kernels even if they were built in different places."""
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for f in [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "pcp_hip.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "pcp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit into pointcloudprocessor_amd/lib/libpcp_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # hipcc runs INSIDE csrc/ on relative file names, one translation unit per command with a FIXED compilation-unit id: the
    # object embeds the names it was given, and hipcc's default unit id is a hash of the source's path that ends up in symbol
    # names (__hip_cuid_<hash>) -- a library built from the same sources must be the same bytes wherever the tree lies
    # (profiles/build_stamp.py keys the counter summaries on its hash).  The units compile side by side.
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    inc = ["-I", os.path.relpath(INCLUDE, CSRC), "-I", "."]

    def compile_one(src):
        stem = os.path.splitext(src)[0]
        obj = os.path.join(os.path.relpath(obj_dir, CSRC), stem + ".o")
        cmd = [_hipcc()] + HIPCC_FLAGS + inc + [f"-cuid={stem}", "-c", src, "-o", obj]
        return cmd, subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    log = ""
    for cmd, proc in results:
        if verbose or proc.returncode != 0:
            log += " ".join(cmd) + "\n" + proc.stdout + proc.stderr
    if any(proc.returncode != 0 for _, proc in results):
        print(log)
        raise RuntimeError("hipcc failed building libpcp_hip.so:\n" + log[-4000:])
    objs = [os.path.join(os.path.relpath(obj_dir, CSRC), os.path.splitext(src)[0] + ".o") for src in SOURCES]
    link = [_hipcc(), "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    proc = subprocess.run(link, capture_output=True, text=True, cwd=CSRC)
    if verbose or proc.returncode != 0:
        print(log + " ".join(link))
        print(proc.stdout)
        print(proc.stderr)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed linking libpcp_hip.so:\n" + proc.stderr[-4000:])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
