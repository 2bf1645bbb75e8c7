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
    "-shared",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    # no compilation-unit id: hipcc derives it from the source's PATH and puts it into symbol names (__hip_cuid_<hash>), which
    # made the library's bytes depend on where the tree lies; nothing here needs one (no static device variables, no -fgpu-rdc)
    "-fuse-cuid=none",
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
    # hipcc runs INSIDE csrc/ with relative file names: the object embeds the names it was given, and a library built from the
    # same sources must be the same bytes wherever the tree lies (profiles/build_stamp.py keys the counter summaries on its hash)
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", os.path.relpath(INCLUDE, CSRC), "-I", ".", "-o", LIB_PATH] + list(SOURCES)
    proc = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if verbose or proc.returncode != 0:
        print(" ".join(cmd))
        print(proc.stdout)
        print(proc.stderr)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed building libpcp_hip.so:\n" + proc.stderr[-4000:])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
