#!/usr/bin/env python3
"""The build a counter summary belongs to.  Every summary under profiles/ that bench.py reads carries
`"_build": {"lib_sha256": ...}` -- the SHA-256 of the libpcp_hip.so the counters were collected on -- and bench.py drops the
figures derived from a summary (and says `"stale": true`) when the library it runs is another one (a
round-3 counter file had been divided by round-4 durations).  hipcc's output is reproducible: the same sources give the
same library, byte for byte, here and on the GPU box.

    python3 profiles/build_stamp.py              prints the stamp of the library in the tree
    python3 profiles/build_stamp.py FILE.json    writes it into a summary (top-level key "_build")
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pointcloudprocessor_amd", "lib", "libpcp_hip.so")


def lib_sha256(path=None):
    path = os.environ.get("PCP_HIP_LIBRARY") or path or LIB
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def source_sha256():
    sys.path.insert(0, ROOT)
    from pointcloudprocessor_amd import _build

    return _build.source_sha256()


def stamp():
    # lib_sha256: the bytes that ran; source_sha256: what they were built from (a library rebuilt elsewhere from the same
    # sources counts as the same build even if a linker detail made its bytes differ)
    return {"lib_sha256": lib_sha256(), "source_sha256": source_sha256()}


def stamp_file(fn):
    with open(fn) as fh:
        d = json.load(fh)
    d["_build"] = stamp()
    with open(fn, "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)


def read(fn, running_sha=None):
    """(summary or {}, stale): stale = the summary names another library than the one running (or none at all)."""
    try:
        with open(fn) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return {}, True
    b = d.get("_build", {})
    fresh = (b.get("lib_sha256") is not None and b.get("lib_sha256") == (running_sha or lib_sha256())) or (
        b.get("source_sha256") is not None and b.get("source_sha256") == source_sha256())
    return d, not fresh


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for f in sys.argv[1:]:
            stamp_file(f)
    print(json.dumps(stamp()))
