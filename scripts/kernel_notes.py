#!/usr/bin/env python3
"""Register / spill / LDS notes of every kernel in the shipped libpcp_hip.so (gfx950 code objects' metadata).

    python scripts/kernel_notes.py [substring ...]      # e.g.  python scripts/kernel_notes.py hpr sor
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pointcloudprocessor_amd", "lib", "libpcp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size",
        "private_segment_fixed_size", "agpr_count")


def notes():
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
        blob = open(fat, "rb").read()
        # every code object of the fat binary is an ELF of its own: cut at the ELF magics
        starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01", blob)]
        for i, st in enumerate(starts):
            end = starts[i + 1] if i + 1 < len(starts) else len(blob)
            co = os.path.join(td, f"co{i}.elf")
            with open(co, "wb") as fh:
                fh.write(blob[st:end])
            r = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True)
            if "amdhsa.kernels" not in r.stdout:
                continue
            cur = {}
            for line in r.stdout.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k in KEYS:
                    cur[k] = int(v)
                elif k == "name" and v.startswith("_Z"):
                    cur["name"] = v
                elif k == "wavefront_size":  # the last key of a kernel's entry
                    out.append(cur)
                    cur = {}
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


if __name__ == "__main__":
    ks = notes()
    names = demangle([k.get("name", "?") for k in ks])
    pats = sys.argv[1:]
    print(f"{'kernel':64s} sgpr spill vgpr spill  agpr   lds scratch")
    for k, nm in sorted(zip(ks, names), key=lambda t: t[1]):
        short = re.sub(r"^void ", "", nm)
        short = re.sub(r"\(.*", "", short).replace("pcp::", "")
        if pats and not any(p in short for p in pats):
            continue
        print(f"{short[:64]:64s} {k.get('sgpr_count', -1):4d} {k.get('sgpr_spill_count', -1):5d} {k.get('vgpr_count', -1):4d} "
              f"{k.get('vgpr_spill_count', -1):5d} {k.get('agpr_count', 0):5d} {k.get('group_segment_fixed_size', 0):5d} "
              f"{k.get('private_segment_fixed_size', 0):7d}")
