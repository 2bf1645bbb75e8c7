#!/usr/bin/env python3
"""Wall-clock probe (NOT a test; moved out of tests/test_sor_gpu.py in round 4 -- a millisecond assertion flakes on a
busy node): a quarter slab of the outlier removal of a SHUFFLED 8 M-point cloud costs about a quarter of the filter
(the queries are dealt out by slabs of the stage's own cell order).  Prints one JSON line.

    python profiles/sor_slab_share_probe.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth  # noqa: E402

x, y, z, _ = synth.make_cloud(8_000_000, seed=4)
perm = np.random.default_rng(5).permutation(len(x))
x, y, z = x[perm].copy(), y[perm].copy(), z[perm].copy()
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
ctx.timing_enable(True)


def sor_ms(slab, slabs):
    ctx.sor_partial(60, slab, slabs)
    ctx.timing_reset()
    ctx.sor_partial(60, slab, slabs)
    return ctx.timing_get(capi.K_SOR)[0]


whole = sor_ms(0, 1)
quarters = [sor_ms(r, 4) for r in range(4)]
print(json.dumps({"whole_ms": round(whole, 3), "quarters_ms": [round(q, 3) for q in quarters],
                  "max_quarter_share": round(max(quarters) / whole, 3), "sum_share": round(sum(quarters) / whole, 3)}))
ctx.close()
