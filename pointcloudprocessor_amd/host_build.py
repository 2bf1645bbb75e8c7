"""g++ build of the C++ host side above the C ABI (pointcloudprocessor_amd/host/)."""
from __future__ import annotations

import os
import subprocess

from . import _build

HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
BIN = os.path.join(HOST, "bin")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

TARGETS = {
    "pcp_shim_selftest": ["shim_selftest.cpp"],
    "PointCloudProcessor": ["main.cpp"],
    "image_dump": ["image_dump.cpp"],
    "format_selftest": ["format_selftest.cpp"],
}


def build(force: bool = False) -> dict:
    os.makedirs(BIN, exist_ok=True)
    out = {}
    for name, srcs in TARGETS.items():
        exe = os.path.join(BIN, name)
        deps = [os.path.join(HOST, s) for s in srcs] + [os.path.join(HOST, "pcp_shim.hpp"), os.path.join(HOST, "pcp_multi.hpp"), os.path.join(HOST, "pcd_io.hpp"), os.path.join(HOST, "image_io.hpp"),
                                                       os.path.join(_build.INCLUDE, "pcp_hip.h")]
        deps = [d for d in deps if os.path.exists(d)]
        stale = force or not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps)
        if stale:
            extra = os.environ.get("PCP_HOST_CXXFLAGS", "").split()  # e.g. -fsanitize=address,undefined (CPU-side checks)
            # the multi-GPU host (pcp_multi.hpp) calls the HIP runtime and RCCL directly: host-only API use, plain g++
            multi = ["-D__HIP_PLATFORM_AMD__", "-isystem", os.path.join(ROCM, "include")] if name == "PointCloudProcessor" else []
            multi_libs = ["-L", os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib")] \
                if name == "PointCloudProcessor" else []
            cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra"] + extra + multi + ["-I", _build.INCLUDE, "-I", HOST] + [
                os.path.join(HOST, s) for s in srcs] + ["-L", _build.LIB_DIR, "-lpcp_hip"] + multi_libs + [
                                                        "-lz", "-Wl,-rpath,$ORIGIN/../../lib", "-o", exe]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(f"g++ failed for {name}:\n{proc.stderr[-3000:]}")
        out[name] = exe
    return out
