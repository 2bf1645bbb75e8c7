#!/usr/bin/env python3
"""Randomised parity stress of the smoothing stages (not collected by pytest): crops of the bench scene at random
places and densities -> MLS (NONE), SOR and, on a thinned crop, VOXEL_GRID_DILATION, the HIP path against the oracle.
    python scripts/stress_smooth.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_capi as oc  # noqa: E402
from pointcloudprocessor_amd import capi, synth  # noqa: E402


def compare_mls(got, ref, R):
    if not np.array_equal(got["index"], ref["index"]):
        return "index sets differ"
    if len(ref["index"]) == 0:
        return None
    d = np.abs(got["xyz"].astype(np.float64) - ref["xyz"].astype(np.float64))
    if d.max() > 1e-4 * R:
        return f"xyz off by {d.max():.3g}"
    sgn = np.sign((got["normal"].astype(np.float64) * ref["normal"]).sum(axis=1))
    sgn[sgn == 0] = 1.0
    dn = np.abs(got["normal"] * sgn[:, None] - ref["normal"]).max()
    if dn > 1e-4:
        return f"normal off by {dn:.3g}"
    if not np.allclose(got["curvature"], ref["curvature"], rtol=1e-4, atol=1e-9):
        return "curvature"
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    oc.build()
    X, Y, Z, _ = synth.make_cloud(6_000_000)
    threads = oc.hardware_threads()
    for case in range(cases):
        rng = np.random.default_rng(500 + case)
        i0 = int(rng.integers(0, len(X)))
        c = np.array([X[i0], Y[i0], Z[i0]])  # crops centred on the surfaces
        half = rng.uniform(0.25, 0.7)
        sel = (np.abs(X - c[0]) < half) & (np.abs(Y - c[1]) < half) & (np.abs(Z - c[2]) < half)
        keep_frac = rng.choice([1.0, 1.0, 0.5, 0.15])
        sel &= rng.random(len(X)) < keep_frac
        x, y, z = X[sel].copy(), Y[sel].copy(), Z[sel].copy()
        if len(x) < 200:
            print(f"case {case:3d} skipped ({len(x)} points)")
            continue
        R = float(rng.choice([0.03, 0.03, 0.02, 0.05]))
        ctx = capi.Context(0)
        ctx.set_camera(capi.default_camera())
        ctx.upload_cloud(x, y, z)
        mp, op = capi.default_mls_params(), oc.default_mls_params()
        for p in (mp, op):
            p.upsampling = 0
            p.search_radius = R
            p.sqr_gauss_param = R * R
        op.threads = threads
        err = compare_mls(ctx.mls_fetch(ctx.mls_process(mp)), oc.mls(x, y, z, op), R)
        # SOR
        k = int(rng.choice([60, 60, 20, 100]))
        mul = float(rng.choice([0.7, 1.0, 0.3]))
        keep_g, kept_g = ctx.sor(k, mul)
        keep_r, kept_r, dist, thr = oc.sor(x, y, z, k, mul, threads=threads, details=True)
        diff = np.nonzero(keep_g != keep_r)[0]
        if not np.all(np.abs(dist[diff] - thr) <= 1e-6 * thr):
            err = err or f"SOR flags differ away from the threshold ({len(diff)})"
        redo = ctx.sor_redo_fraction()
        # voxel dilation on a thin slice
        vs = (np.abs(x - x.mean()) < 0.04) & (rng.random(len(x)) < 0.5)
        nv = 0
        if 50 < vs.sum() < 4000:
            ctx.upload_cloud(x[vs], y[vs], z[vs])
            vp, vo = capi.default_mls_params(), oc.default_mls_params()
            for p in (vp, vo):
                p.vgd_voxel_size = 0.003
                p.vgd_iterations = 2
            vo.threads = threads
            gv = ctx.mls_fetch(ctx.mls_process(vp))
            rv = oc.mls_voxel_dilation(x[vs], y[vs], z[vs], vo)
            nv = len(rv["index"])
            err = err or compare_mls(gv, rv, 0.03)
            # the whole chain on the slice, one-shot against streamed (round 5): the same rows, bit for bit -- with a random
            # first guess of the halo now and then (a guess too small is redone wider)
            vp.sor_mean_k = int(rng.choice([60, 20]))
            one = ctx.mls_fetch(ctx.cloud_smooth(vp))
            halo = rng.choice([None, None, "2", "6"])
            if halo:
                os.environ["PCP_CSS_HALO"] = halo
            cap = 8192
            while True:
                try:
                    total, kept, chunks = ctx.cloud_smooth_stream_begin(vp, cap)
                    break
                except capi.PcpError as e:  # a plane of the slice holds more voxels than the chunk may
                    if e.code != capi.PCP_ERR_RANGE or cap > (1 << 24):
                        raise
                    cap *= 4
            os.environ.pop("PCP_CSS_HALO", None)
            parts = []
            while True:
                mchunk = ctx.cloud_smooth_stream_next()
                if mchunk == 0:
                    break
                parts.append(ctx.mls_fetch(mchunk))
            for key in ("index", "xyz", "normal", "curvature"):
                got = np.concatenate([q[key] for q in parts]) if parts else one[key][:0]
                if not np.array_equal(got, one[key]):
                    err = err or f"streamed chain differs from the one-shot chain in {key} ({chunks} chunks, halo {halo})"
            if kept != len(one["index"]):
                err = err or f"streamed chain kept {kept}, one-shot {len(one['index'])}"
        ctx.close()
        print(f"case {case:3d} n={len(x):7d} r={R:.2f} k={k:3d} mul={mul:.1f} sor_redo={redo:.3f} voxels={nv:7d}  "
              f"{'ok' if not err else 'MISMATCH: ' + err}", flush=True)
        if err:
            sys.exit(1)
    print("all cases within tolerance")


if __name__ == "__main__":
    main()
