"""world_size-2 `gloo` rehearsal of the multi-GPU path: points are
sharded by contiguous index range, the only exchange on the data path is an
all-reduce(MIN) of the per-keyframe depth maps, outputs are all-gathered.  The
driver under test is pointcloudprocessor_amd.pipeline.PointCloudColorizer; the
compute engine here is an oracle-backed stand-in defined in this test (the
product has no CPU engine)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same hooks as pipeline.HipEngine, computed with the numpy twin."""

    def __init__(self, cam, x, y, z, poses, images):
        self.cam, self.x, self.y, self.z, self.poses, self.images = cam, x, y, z, poses, images
        self.depth = None
        self.proj = None

    @property
    def n_frames(self):
        return len(self.poses)

    def depth_pass(self, f0=0, f1=None):
        from oracle import np_oracle as npo

        f1 = len(self.poses) if f1 is None else f1
        self._alloc()
        cells = self.depth.size // len(self.poses)
        for f in range(f0, f1):
            w2c, _ = npo.pose_to_matrices(self.poses[f])
            _, dmap, p = npo.cull_frame(self.cam, w2c, self.x, self.y, self.z)
            self.depth[f * cells:(f + 1) * cells] = dmap.reshape(-1)
            self.proj[f] = p

    def _alloc(self):
        if self.depth is None:
            mw, mh = self.cam["cull_width"] // 14, self.cam["cull_height"] // 14
            self.depth = np.zeros(len(self.poses) * mw * mh, np.float32)
            self.proj = [None] * len(self.poses)

    def depth_maps_tensor(self):
        import torch

        self._alloc()
        return torch.from_numpy(self.depth)  # shares memory: the all-reduce lands in self.depth

    def colour_from_depth(self, download=True):
        from oracle import np_oracle as npo

        n = len(self.x)
        cells = self.depth.size // len(self.poses)
        lists = [[] for _ in range(n)]
        for f, pose in enumerate(self.poses):
            p = self.proj[f]
            dmap = self.depth[f * cells:(f + 1) * cells]
            inmap = p["cell"] >= 0
            keep = np.zeros(n, bool)
            keep[inmap] = ~(p["range64"][inmap] > dmap[p["cell"][inmap]].astype(np.float64) + 0.05)
            sel = np.nonzero(keep & (p["pixel"] >= 0))[0]
            if len(sel) == 0:
                continue
            bgr = np.asarray(self.images[f]).reshape(-1, 3)[p["pixel"][sel]]
            _, _, fin = npo.scores(p["xc"][sel], p["yc"][sel], p["zc"][sel], pose)
            for k, i in enumerate(sel):
                lists[i].append((float(fin[k]), int(bgr[k, 2]), int(bgr[k, 1]), int(bgr[k, 0])))
        rgb = np.zeros((n, 3), np.uint8)
        for i, lst in enumerate(lists):
            if not lst:
                continue
            lst = sorted(lst, key=lambda e: -e[0])[:5]
            tot = np.float32(0)
            acc = [np.float32(0)] * 3
            for s, r, g, b in lst:
                s = np.float32(s)
                acc = [acc[0] + np.float32(r) * s, acc[1] + np.float32(g) * s, acc[2] + np.float32(b) * s]
                tot = tot + s
            rgb[i] = [np.uint8(int(a / tot)) for a in acc]
        return dict(rgb=rgb, has=(rgb != 0).any(axis=1).astype(np.uint8))


class OracleMlsCtx:
    """mls_process_slab / mls_fetch of capi.Context, computed with the C oracle."""

    def __init__(self, x, y, z):
        self.xyz = (x, y, z)
        self.last = None

    def mls_process_slab(self, params, slab, n_slabs):
        """a slab of a spatial order of the queries (here: sorted by x): every point belongs to exactly one slab"""
        from oracle import oracle_capi as oc

        op = oc.default_mls_params()
        op.upsampling = 0
        op.threads = 2
        r = oc.mls(*self.xyz, op)
        n = len(self.xyz[0])
        place = np.empty(n, np.int64)
        place[np.argsort(self.xyz[0], kind="stable")] = np.arange(n)
        lo, hi = n * slab // n_slabs, n * (slab + 1) // n_slabs
        sel = (place[r["index"]] >= lo) & (place[r["index"]] < hi)
        self.last = {k: v[sel] for k, v in r.items()}
        return int(sel.sum())

    def mls_fetch(self, m):
        return self.last


class OracleMlsEngine:
    def __init__(self, x, y, z):
        self.ctx = OracleMlsCtx(x, y, z)


class OracleSorCtx:
    """sor_chunk_points / sor_partial / sor_finish of capi.Context on the C oracle's mean distances: the protocol of
    pcp_sor_partial / pcp_sor_finish (slabs of a spatial order -- here: sorted by x --, per-chunk sums in a fixed order,
    threshold from the array put together), with a small chunk so that two ranks own several chunks each."""

    CHUNK = 256

    def __init__(self, x, y, z, mean_k):
        from oracle import oracle_capi as oc

        _, _, self.dist, _ = oc.sor(x, y, z, mean_k, 1.0, threads=2, details=True)
        self.dist = self.dist.astype(np.float32)
        self.n = len(x)
        self.order = np.argsort(x, kind="stable")  # place -> index

    def sor_chunk_points(self):
        return self.CHUNK

    @staticmethod
    def chunk_sums(d):
        d32 = d.astype(np.float32)
        sq = (d32 * d32).astype(np.float64)  # fp32 squares, as the reference
        return np.array([np.sum(d32.astype(np.float64)), np.sum(sq)])

    def _slab(self, slab, n_slabs):
        chunks = (self.n + self.CHUNK - 1) // self.CHUNK
        c0, c1 = chunks * slab // n_slabs, chunks * (slab + 1) // n_slabs
        return c0, c1

    def sor_partial(self, mean_k, slab, n_slabs):
        c0, c1 = self._slab(slab, n_slabs)
        c = self.CHUNK
        sums = [self.chunk_sums(self.dist[self.order[k * c:min((k + 1) * c, self.n)]]) for k in range(c0, c1)]
        return c0, np.array(sums).reshape(-1, 2)

    def sor_finish(self, std_mul, sums, slab, n_slabs):
        assert len(sums) == (self.n + self.CHUNK - 1) // self.CHUNK
        s, q = 0.0, 0.0
        for a, b in np.asarray(sums, np.float64):  # fixed order
            s, q = s + a, q + b
        n = float(self.n)
        thr = s / n + std_mul * np.sqrt((q - s * s / n) / (n - 1.0))
        c0, c1 = self._slab(slab, n_slabs)
        mine = self.order[c0 * self.CHUNK:min(c1 * self.CHUNK, self.n)]
        keep = np.zeros(self.n, np.uint8)
        keep[mine] = ~(self.dist[mine].astype(np.float64) > thr)
        return keep, int(keep.sum())


def _sor_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from pointcloudprocessor_amd import pipeline
    from test_sharding_gloo import OracleMlsEngine, OracleSorCtx, _mls_points

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, z = _mls_points()

    class P:
        sor_mean_k, sor_std_mul = 12, 0.7

    smooth = pipeline.CloudSmooth.__new__(pipeline.CloudSmooth)
    smooth.engine = OracleMlsEngine(x, y, z)
    smooth.engine.ctx = OracleSorCtx(x, y, z, P.sor_mean_k)
    smooth.params = P
    keep = smooth.outlier_removal_sharded(len(x), rank, world)
    np.save(os.path.join(out_dir, f"sor{rank}.npy"), keep)
    dist.barrier()
    dist.destroy_process_group()


def _mls_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from pointcloudprocessor_amd import pipeline
    from test_sharding_gloo import OracleMlsEngine, _mls_points

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, z = _mls_points()
    smooth = pipeline.CloudSmooth.__new__(pipeline.CloudSmooth)
    smooth.engine = OracleMlsEngine(x, y, z)
    smooth.params = None
    full = smooth.process_sharded(len(x), rank, world)
    np.savez(os.path.join(out_dir, f"mls{rank}.npz"), **full)
    dist.barrier()
    dist.destroy_process_group()


def _mls_points():
    rng = np.random.default_rng(8)
    a = rng.uniform(-0.1, 0.1, (1500, 2))
    z = 0.5 * a[:, 0] ** 2 + rng.normal(0, 5e-4, 1500)
    stray = rng.uniform(-3, 3, (7, 3))
    pts = np.concatenate([np.stack([a[:, 0], a[:, 1], z], 1), stray]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    return pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from pointcloudprocessor_amd import pipeline, synth
    from test_sharding_gloo import OracleEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam = synth.camera_dict("tiny")
    x, y, z, _ = synth.make_cloud(n, seed=77)
    poses, _ = synth.make_trajectory(4)
    imgs = [synth.make_image(f, cam["image_width"], cam["image_height"]) for f in range(4)]
    lo, hi = pipeline.shard_bounds(n, rank, world)
    eng = OracleEngine(cam, x[lo:hi], y[lo:hi], z[lo:hi], poses, imgs)
    col = pipeline.PointCloudColorizer(eng, rank, world)
    local = col.run()
    full = col.gather(local, n)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rgb=full["rgb"], has=full["has"], depth=eng.depth)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    from pointcloudprocessor_amd import pipeline

    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            b = [pipeline.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp

    from conftest import cam_struct
    from pointcloudprocessor_amd import synth

    n, world = 6000, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    cam = synth.camera_dict("tiny")
    x, y, z, _ = synth.make_cloud(n, seed=77)
    poses, _ = synth.make_trajectory(4)
    imgs = [synth.make_image(f, cam["image_width"], cam["image_height"]) for f in range(4)]
    cp = oracle.default_cull_params()
    cp.match_mode = oracle.MATCH_IDENTITY  # the stand-in engine above scores the transform output (npo.scores)
    ref = oracle.colorize(cam_struct(oracle, cam), cp, x, y, z, poses, imgs)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # the reduced depth maps are identical on both ranks and equal the single-process maps
    assert np.array_equal(r0["depth"], r1["depth"])
    for f in range(4):
        w2c, _ = oracle.pose_to_matrices(poses[f])
        _, dmap, _ = oracle.cull_frame(cam_struct(oracle, cam), oracle.default_cull_params(), w2c, x, y, z)
        cells = dmap.size
        assert np.array_equal(r0["depth"][f * cells:(f + 1) * cells], dmap.reshape(-1))
    # sharded colours == unsharded colours, bit for bit, on every rank
    for r in (r0, r1):
        assert np.array_equal(r["rgb"], ref["rgb"]) and np.array_equal(r["has"], ref["has"])
    assert ref["has"].sum() > 100


@pytest.mark.timeout(300)
def test_two_rank_outlier_removal_sharding(tmp_path, oracle):
    """StatisticalOutlierRemoval with the queries dealt out by slabs of a spatial order over 2 gloo ranks
    (pipeline.CloudSmooth.outlier_removal_sharded: all-reduce of the chunk sums, every rank classifies its range, all-reduce of
    the flags): the keep mask equals the one-process filter on every rank."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_sor_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y, z = _mls_points()
    ctx = OracleSorCtx(x, y, z, 12)
    _, sums = ctx.sor_partial(12, 0, 1)
    ref, kept = ctx.sor_finish(0.7, sums, 0, 1)
    keep_o, kept_o = oracle.sor(x, y, z, 12, 0.7, threads=2)[:2]
    assert 0 < kept < len(x) and abs(kept - kept_o) <= 2  # the chunked sums against PCL's running sums: a borderline point at most
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"sor{r}.npy"), ref), r


@pytest.mark.timeout(300)
def test_two_rank_mls_query_sharding(tmp_path, oracle):
    """MLS with queries dealt out by slabs of a spatial order over 2 gloo ranks: the all-gathered result, merged by source
    index, equals the single-process result on every rank."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_mls_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y, z = _mls_points()
    op = oracle.default_mls_params()
    op.upsampling = 0
    op.threads = 2
    ref = oracle.mls(x, y, z, op)
    assert 0 < len(ref["index"]) < len(x)
    for r in range(2):
        got = np.load(tmp_path / f"mls{r}.npz")
        for k in ("index", "xyz", "normal", "curvature"):
            assert np.array_equal(got[k], ref[k]), (r, k)


# ---- hidden_points_removal over index shards: the exchange of the verdicts (pipeline.HullSharding) ----
def _hull_flags(f, n):
    """stand-in verdicts of keyframe f for the n points of the map (the protocol under test moves them, it does not compute them)"""
    return (((np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97 * f + 13)) >> np.uint64(9)) & np.uint64(1)).astype(np.uint8)


class _FakeHullCtx:
    def __init__(self, n):
        self.n, self.done = n, set()

    def depth_pass(self, f0, f1):
        self.done |= set(range(f0, f1))

    def synchronize(self):
        pass

    def cull_frame(self, f):
        assert f in self.done, "a keyframe outside this rank's block was asked for"
        keep = _hull_flags(f, self.n)
        return keep, None, int(keep.sum())


class _FakeShardCtx:
    def __init__(self):
        self.got = {}

    def hull_flags_import(self, f, piece):
        assert f not in self.got
        self.got[f] = np.array(piece, np.uint8)

    def synchronize(self):
        pass


def _hull_worker(rank, world, port, out_dir, n, n_frames):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from pointcloudprocessor_amd import pipeline
    from test_sharding_gloo import _FakeHullCtx, _FakeShardCtx

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = _FakeShardCtx()
    hs = pipeline.HullSharding(_FakeHullCtx(n), shard, n, rank, world)
    res = hs.run(n_frames)
    np.savez(os.path.join(out_dir, f"hull{rank}.npz"), kept=res["kept"], frames=np.array(sorted(shard.got)),
             exchange_bytes=res["exchange_bytes"], rounds=res["rounds"], **{f"f{f}": v for f, v in shard.got.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 4)])
def test_hull_verdicts_reach_every_index_shard(tmp_path, world, n_frames):
    """Every rank takes the hulls of its block of keyframes on a whole-map context; after the exchange every rank holds,
    for EVERY keyframe, the verdicts of exactly its own index range (uneven blocks and uneven shards included)."""
    import torch.multiprocessing as mp

    from pointcloudprocessor_amd import pipeline

    n = 1001
    port = _free_port()
    mp.spawn(_hull_worker, args=(world, port, str(tmp_path), n, n_frames), nprocs=world, join=True)
    kept = 0
    for r in range(world):
        d = np.load(tmp_path / f"hull{r}.npz")
        lo, hi = pipeline.shard_bounds(n, r, world)
        assert list(d["frames"]) == list(range(n_frames))
        for f in range(n_frames):
            assert np.array_equal(d[f"f{f}"], _hull_flags(f, n)[lo:hi]), (r, f)
        kept += int(d["kept"])
        # what a rank moves per round: the map's verdicts ONE BIT per point, each slice padded to whole bytes -- not the
        # W x n bytes of an all-gather of byte flags
        assert int(d["rounds"]) == max(b - a for a, b in (pipeline.keyframe_block(n_frames, k, world) for k in range(world)))
        assert int(d["exchange_bytes"]) <= int(d["rounds"]) * (n // 8 + world)
    assert kept == sum(int(_hull_flags(f, n).sum()) for f in range(n_frames))
    blocks = [pipeline.keyframe_block(n_frames, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
