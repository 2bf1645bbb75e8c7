"""CPU simulation (numpy, scipy's hull as the truth) of what the quick hidden certificate of pcp_hpr.hip can settle on a real
candidate set of C3: the four triangles of neighbouring cells, all 56 triangles of the ring, with the own cell's outermost
candidate, the fan around it, triangles two cells away.  python profiles/hpr_quick_sim.py [points] [keyframe]"""
import sys, time, itertools
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import ConvexHull
from oracle import np_oracle as no
from pointcloudprocessor_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kf = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(256)
w2c, _ = no.pose_to_matrices(poses[kf])
p = no.project_frame(cam, w2c, x, y, z)
u, v = p["u"], p["v"]
ok = (p["zc"] > 0) & no._trunc_ok(u) & no._trunc_ok(v)
ui = np.where(ok, np.trunc(np.where(ok, u, 0)), -1); vi = np.where(ok, np.trunc(np.where(ok, v, 0)), -1)
cand = ok & (ui >= 0) & (ui < cam["cull_width"]) & (vi >= 0) & (vi < cam["cull_height"])
idx = np.nonzero(cand)[0]
pts = np.stack([p["xc"][idx], p["yc"][idx], p["zc"][idx]], axis=1).astype(np.float64)
nrm = np.linalg.norm(pts, axis=1)[:, None]
F = pts + (2.0 * (90000.0 - nrm) * pts) / nrm
m = len(F)
t0 = time.time()
hull = ConvexHull(np.concatenate([F, np.zeros((1, 3))]))
vis = np.zeros(m, bool); vis[[k for k in hull.vertices if k < m]] = True
print("candidates", m, "visible", vis.sum(), "hull s", round(time.time() - t0, 1), flush=True)
a, b = F[:, 0] / F[:, 2], F[:, 1] / F[:, 2]
rho = np.linalg.norm(F, axis=1)
wa, wb = a.max() - a.min(), b.max() - b.min()
def run(target):
    h = np.sqrt(target * wa * wb / m)
    gw, gh = int(wa / h) + 1, int(wb / h) + 1
    ci = np.minimum(gw - 1, ((a - a.min()) / h).astype(int)); cj = np.minimum(gh - 1, ((b - b.min()) / h).astype(int))
    cell = cj * gw + ci
    order = np.lexsort((rho, cell))
    rep = np.full(gw * gh, -1); rep[cell[order]] = order   # last = largest rho per cell
    hid = np.nonzero(~vis)[0]
    P = F[hid]
    def reps(di, dj):
        i, j = ci[hid] + di, cj[hid] + dj
        okk = (i >= 0) & (j >= 0) & (i < gw) & (j < gh)
        r = np.where(okk, rep[np.where(okk, j * gw + i, 0)], -1)
        r = np.where(r == hid, -1, r)
        return r
    def det(A, B, C):
        return np.einsum('ij,ij->i', A, np.cross(B, C))
    def inside(ra, rb, rc):
        okk = (ra >= 0) & (rb >= 0) & (rc >= 0)
        A, B, C = F[np.maximum(ra, 0)], F[np.maximum(rb, 0)], F[np.maximum(rc, 0)]
        # p in tetra (0,A,B,C): same sign of det for (A,B,C) orientation with p replacing each, and p below plane
        d = det(A, B, C)
        d1, d2, d3 = det(P, B, C), det(A, P, C), det(A, B, P)
        s = np.sign(d)
        cone = (np.sign(d1) == s) & (np.sign(d2) == s) & (np.sign(d3) == s) & (s != 0)
        # below plane: orient(A,B,C,P) same side as origin
        n = np.cross(B - A, C - A)
        sp = np.einsum('ij,ij->i', n, P - A); so = np.einsum('ij,ij->i', n, -A)
        return okk & cone & (np.sign(sp) == np.sign(so)) & (sp != 0)
    nb = {(di, dj): reps(di, dj) for di in (-1, 0, 1) for dj in (-1, 0, 1)}
    cur = [((-1,-1),(1,-1),(0,1)), ((-1,1),(1,1),(0,-1)), ((-1,-1),(-1,1),(1,0)), ((1,-1),(1,1),(-1,0))]
    hitA = np.zeros(len(hid), bool)
    for t in cur: hitA |= inside(nb[t[0]], nb[t[1]], nb[t[2]])
    keys8 = [k for k in nb if k != (0, 0)]
    hitB = np.zeros(len(hid), bool)
    for t in itertools.combinations(keys8, 3): hitB |= inside(nb[t[0]], nb[t[1]], nb[t[2]])
    hitC = hitB.copy()
    for t in itertools.combinations(keys8, 2): hitC |= inside(nb[(0, 0)], nb[t[0]], nb[t[1]])
    # ring of distance 2 (16 cells) corners/mids only: 8 reps
    nb2 = {(di, dj): reps(di, dj) for di in (-2, 0, 2) for dj in (-2, 0, 2) if (di, dj) != (0, 0)}
    hitD = hitA.copy()
    for t in cur:
        hitD |= inside(nb2[(2*t[0][0], 2*t[0][1])], nb2[(2*t[1][0], 2*t[1][1])], nb2[(2*t[2][0], 2*t[2][1])])
    ring = [(-1,-1),(0,-1),(1,-1),(1,0),(1,1),(0,1),(-1,1),(-1,0)]
    hitE = np.zeros(len(hid), bool)
    for k in range(8): hitE |= inside(nb[(0,0)], nb[ring[k]], nb[ring[(k+1)%8]])
    hitF = hitA | hitE
    hitG = hitF.copy()
    for k in range(8): hitG |= inside(nb[(0,0)], nb[ring[k]], nb[ring[(k+2)%8]])
    print(f"   E(fan of 8 around own rep) {hitE.mean():.3f}  F(A+E) {hitF.mean():.3f}  G(F + fan skipping one) {hitG.mean():.3f}")
    print(f"target {target}: hidden {len(hid)}  A(4 tri) {hitA.mean():.3f}  B(56 tri of 8 nbrs) {hitB.mean():.3f}  C(+own rep) {hitC.mean():.3f}  D(A + 4 tri at distance 2) {hitD.mean():.3f}", flush=True)
for tg in (8.0,):
    run(tg)
