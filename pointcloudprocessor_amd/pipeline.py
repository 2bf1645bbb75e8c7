"""Host-side mirror of the reference's operator interface for the hot path, on
top of the C ABI (capi.py).  Names follow the reference:

  ViewCulling.cull            vlcal::ViewCulling::cull     (view_culling.hpp:34)
  PointCloudColorizer.run     pcdColorizationAndSmooth     (PointCloudProcessor.cpp:474-602)
  CloudSmooth.process         CloudSmooth::process         (cloudSmooth.cpp:77-185)

plus the point-index sharding of SURVEY.md section 8(e): every rank owns a
contiguous slice of the map, all keyframes / images are replicated, and the one
exchange step is an all-reduce(MIN) of the per-keyframe depth maps.  Per-point
results (top-5 lists, colours) are rank-local, so no other collective is on the
data path; outputs are assembled with an all-gather when asked for.

The compute engine is libpcp_hip.so; there is no CPU engine in this package.
"""
from __future__ import annotations

import numpy as np

from . import capi


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous index range of `rank` (first n % world ranks get one extra point)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _DeviceArray:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr: int, n: int, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipEngine:
    """One GPU worth of the hot path (one pcp_context)."""

    def __init__(self, device: int = 0):
        self.ctx = capi.Context(device)
        self.device = device

    def configure(self, camera: dict | capi.Camera, cull: capi.CullParams | None = None):
        cam = camera if isinstance(camera, capi.Camera) else capi.camera_from_dict(camera)
        self.ctx.set_camera(cam, cull)

    def upload_cloud(self, x, y, z):
        self.ctx.upload_cloud(x, y, z)

    def set_keyframes(self, poses, images=None, masks=None, T_opt=None):
        self.ctx.set_frames(poses, T_opt)
        if images is not None:
            for f, im in enumerate(images):
                self.ctx.upload_image(f, im)
        if masks is not None:
            for f, m in enumerate(masks):
                if m is not None:
                    self.ctx.upload_mask(f, m)

    def use_torch_stream(self):
        """Run the library's kernels on torch's current HIP stream: the collective is then
        stream-ordered against them (RCCL waits on / is waited for through events), with no
        host synchronisation between the depth pass, the all-reduce and the colour pass."""
        import torch

        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._on_torch_stream = True

    # hooks used by the sharded driver
    @property
    def n_frames(self):
        return self.ctx.n_frames

    def depth_pass(self, f0: int = 0, f1: int | None = None):
        self.ctx.depth_pass(f0, f1)

    def depth_maps_tensor(self):
        """torch view (no copy) of the device-resident depth maps, for RCCL."""
        import torch

        ptr, n = self.ctx.depth_maps_device()
        if not getattr(self, "_on_torch_stream", False):
            self.ctx.synchronize()
        # the view is rebuilt whenever the library may have reallocated or re-shaped the maps (new address, length,
        # keyframe count or cull geometry); otherwise one wrapper per step would cost ~50 us of host time
        key = (ptr, n, self.ctx.n_frames, self.ctx.map_shape)
        cached = getattr(self, "_depth_tensor", None)
        if cached is None or cached[0] != key:
            self._depth_tensor = (key, torch.as_tensor(_DeviceArray(ptr, n), device=f"cuda:{self.device}"))
        return self._depth_tensor[1]

    def colour_from_depth(self, download=True):
        return self.ctx.colorize_from_depth(download=download)

    def close(self):
        self.ctx.close()


class ViewCulling:
    """vlcal::ViewCulling with the z-buffer routine (view_culling.cpp:52-174)."""

    def __init__(self, engine: HipEngine):
        self.engine = engine

    def cull(self, keyframe: int):
        """Returns (indices kept, in input order; depth map)."""
        keep, dmap, _ = self.engine.ctx.cull_frame(keyframe)
        return np.nonzero(keep)[0].astype(np.int32), dmap


class PointCloudColorizer:
    """pcdColorizationAndSmooth over a (possibly sharded) map."""

    def __init__(self, engine, rank: int = 0, world: int = 1, group=None, chunks: int = 0):
        self.engine = engine
        self.rank = rank
        self.world = world
        self.group = group
        self.chunks = chunks

    def run(self, download: bool = True):
        """Local points' colours: dict(rgb (n,3) uint8, has (n,) uint8).

        Multi-rank: the keyframes are split into `chunks` groups (0 = chosen from the size of the maps); the
        all-reduce(MIN) of one group's depth maps (RCCL stream) overlaps the depth pass of the next group."""
        if self.world == 1:
            self.engine.depth_pass()
            return self.engine.colour_from_depth(download=download)
        import torch.distributed as dist

        F = self.engine.n_frames
        t = self.engine.depth_maps_tensor()
        cells = t.numel() // max(F, 1)
        # chunks = 0: one all-reduce for small maps (splitting the depth pass costs ~0.1 ms per extra chunk at C3,
        # more than overlapping a ~10 MB all-reduce saves), two keyframe groups from 32 MB of maps up
        chunks = self.chunks if self.chunks > 0 else (1 if t.numel() * 4 < (32 << 20) else 2)
        # group boundaries on multiples of 32 keyframes (whole tile-mask words) when there are enough keyframes
        align = 32 if F >= 64 * chunks else 1
        bounds = sorted({min(F, ((F * c) // chunks + align - 1) // align * align) for c in range(chunks)} | {0, F})
        works = []
        for f0, f1 in zip(bounds[:-1], bounds[1:]):
            self.engine.depth_pass(f0, f1)
            if not getattr(self.engine, "_on_torch_stream", False) and t.is_cuda:
                self.engine.ctx.synchronize()
            # ranges are positive finite floats: float MIN == the uint-bits MIN the kernel used
            works.append(dist.all_reduce(t[f0 * cells:f1 * cells], op=dist.ReduceOp.MIN, group=self.group, async_op=True))
        for w in works:
            w.wait()  # stream-level wait on GPUs, blocking on gloo
        if t.is_cuda and not getattr(self.engine, "_on_torch_stream", False):
            import torch

            torch.cuda.current_stream().synchronize()
        return self.engine.colour_from_depth(download=download)

    def gather(self, local: dict, n_total: int):
        """All-gather the per-shard colours into full-length arrays (every rank)."""
        if self.world == 1:
            return local
        import torch
        import torch.distributed as dist

        sizes = [shard_bounds(n_total, r, self.world) for r in range(self.world)]
        maxn = max(hi - lo for lo, hi in sizes)
        packed = np.zeros((maxn, 4), np.uint8)
        lo, hi = sizes[self.rank]
        packed[: hi - lo, :3] = local["rgb"]
        packed[: hi - lo, 3] = local["has"]
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        mine = torch.from_numpy(packed).to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        rgb = np.zeros((n_total, 3), np.uint8)
        has = np.zeros(n_total, np.uint8)
        for r, (lo, hi) in enumerate(sizes):
            a = outs[r].cpu().numpy()
            rgb[lo:hi] = a[: hi - lo, :3]
            has[lo:hi] = a[: hi - lo, 3]
        return dict(rgb=rgb, has=has)


def keyframe_block(n_frames: int, rank: int, world: int):
    """[f0, f1): the keyframes whose hulls rank `rank` takes when hidden_points_removal runs over `world` GPUs."""
    return n_frames * rank // world, n_frames * (rank + 1) // world


class HullSharding:
    """hidden_points_removal (view_culling.cpp:266-334) over point-index shards.  A keyframe's hull is taken over EVERY
    candidate of the map, so an index shard cannot decide its own points: every rank holds a second context with the whole
    map (`hull_ctx`, cull_mode PCP_CULL_HPR, no images), takes the hulls of its block of keyframes there in one
    pcp_depth_pass (several keyframes in flight), and the verdicts -- one flag per map point -- go to the ranks that own the
    points: in round i rank o holds keyframe f0(o) + i and sends every rank k the slice [lo(k), hi(k)) of its verdicts, ONE
    BIT per point, in one all_to_all_single (a rank receives W slices of its own index range: ~n / 8 bytes per round and
    rank where an all-gather of byte flags moved W x n to everyone; host/pcp_multi.hpp sends slices the same way).  Every
    rank imports the slices into its shard context (`shard_ctx`, PCP_DEPTH_BATCHED), which then colours / dumps from the
    bits exactly as a one-GPU run does.  With the "nccl" backend the flags stay in device memory from pcp_cull_frame to
    pcp_hull_flags_import (ABI v5)."""

    def __init__(self, hull_ctx, shard_ctx, n_total: int, rank: int, world: int, group=None):
        self.hull_ctx, self.shard_ctx = hull_ctx, shard_ctx
        self.n_total, self.rank, self.world, self.group = int(n_total), rank, world, group

    @staticmethod
    def packed_bytes(count: int) -> int:
        """bytes of a bit-packed slice of `count` verdicts (bit j of byte b = point 8 b + j of the slice)"""
        return (int(count) + 7) // 8

    def run(self, n_frames: int, device: str | None = None):
        """Returns dict(hull_s, exchange_s, kept, exchange_bytes, rounds) -- kept = hull vertices of this rank's keyframes,
        exchange_bytes = what this rank sent (== what it received up to padding) over all rounds."""
        import time

        import torch
        import torch.distributed as dist

        W, r, n = self.world, self.rank, self.n_total
        bounds = [shard_bounds(n, k, W) for k in range(W)]
        lo, hi = bounds[r]
        mine_n = hi - lo
        f0, f1 = keyframe_block(n_frames, r, W)
        t0 = time.perf_counter()
        if f1 > f0:
            self.hull_ctx.depth_pass(f0, f1)
        self.hull_ctx.synchronize()
        t_hull = time.perf_counter() - t0
        on_gpu = W > 1 and dist.get_backend(self.group) == "nccl"
        dev = (device or "cuda") if on_gpu else "cpu"
        blocks = [keyframe_block(n_frames, k, W) for k in range(W)]
        rounds = max(b1 - b0 for b0, b1 in blocks)
        send_sizes = [self.packed_bytes(b - a) for a, b in bounds]      # to rank k: its index range, bit-packed
        recv_sizes = [self.packed_bytes(mine_n)] * W                    # from every rank: my index range
        flags = torch.zeros(n, dtype=torch.uint8, device=dev)
        padded = torch.zeros(8 * max(send_sizes), dtype=torch.uint8, device=dev)  # a slice, zero-padded to whole bytes
        weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
        shifts = torch.arange(8, dtype=torch.uint8, device=dev)
        send = torch.zeros(sum(send_sizes), dtype=torch.uint8, device=dev)
        recv = torch.zeros(sum(recv_sizes), dtype=torch.uint8, device=dev)
        kept = moved = 0
        t0 = time.perf_counter()
        for i in range(rounds):
            f = f0 + i
            if f < f1:
                if on_gpu:
                    kept += self.hull_ctx.cull_frame_into(f, flags.data_ptr())  # from the whole-run bits: nothing recomputed
                else:
                    keep, _, k = self.hull_ctx.cull_frame(f)
                    flags.copy_(torch.from_numpy(np.ascontiguousarray(keep)))
                    kept += int(k)
                at = 0
                for (a, b), sz in zip(bounds, send_sizes):
                    padded.zero_()
                    padded[: b - a] = flags[a:b] != 0
                    send[at:at + sz] = (padded[: 8 * sz].view(sz, 8) * weights).sum(dim=1, dtype=torch.uint8)
                    at += sz
            else:
                send.zero_()  # (no keyframe of mine in this round: the others ignore what arrives from me)
            if W > 1:
                dist.all_to_all_single(recv, send, output_split_sizes=recv_sizes, input_split_sizes=send_sizes, group=self.group)
                moved += int(send.numel())
            else:
                recv.copy_(send)
            psz = recv_sizes[0]
            for k in range(W):
                fk0, fk1 = blocks[k]
                if fk0 + i >= fk1:
                    continue
                bits = ((recv[k * psz:(k + 1) * psz].unsqueeze(1) >> shifts) & 1).reshape(-1)[:mine_n].contiguous()
                if on_gpu:
                    torch.cuda.current_stream().synchronize()  # the library imports on its own stream
                    self.shard_ctx.hull_flags_import_ptr(fk0 + i, bits.data_ptr())
                    self.shard_ctx.synchronize()  # `bits` is released next
                else:
                    self.shard_ctx.hull_flags_import(fk0 + i, bits.cpu().numpy())
        if on_gpu:
            self.shard_ctx.synchronize()
        return dict(hull_s=t_hull, exchange_s=time.perf_counter() - t0, kept=kept, exchange_bytes=moved, rounds=rounds)


class VisualLiDARCalibration:
    """vlcal::VisualLiDARCalibration::calibrate (calibrate.cpp:42-126) over an index-sharded map.

    A keyframe's joint histogram is a sum over its points: every rank accumulates the histograms of its shard, the ranks
    add them (all-reduce SUM, 8 B x F x (7 bins^2 + bins): 0.9 MB at 64 keyframes) and turn the sums into cost and
    gradient -- identical numbers everywhere, so the BFGS loops of all ranks walk in lockstep without a broadcast.  The
    shards' single-keyframe culls need the MIN-merged depth maps (PointCloudColorizer.run or depth pass + all-reduce
    first, then ctx.set_depth_source(True))."""

    def __init__(self, engine: HipEngine, rank: int = 0, world: int = 1, group=None):
        self.engine = engine
        self.rank = rank
        self.world = world
        self.group = group

    def prepare(self) -> int:
        return self.engine.ctx.nid_prepare()

    def _hist_tensor(self):
        import torch

        ptr, n = self.engine.ctx.nid_histograms_device()
        return torch.as_tensor(_DeviceArray(ptr, n, "<f8"), device=f"cuda:{self.engine.device}")

    def evaluate(self, T, bins: int = 16):
        ctx = self.engine.ctx
        ctx.nid_accumulate(T, bins)
        if self.world > 1:
            import torch.distributed as dist

            ctx.synchronize()
            t = self._hist_tensor()
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                import torch

                torch.cuda.current_stream().synchronize()
            else:  # gloo rehearsal: through the host
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
                import torch

                torch.cuda.synchronize()
        cost, grad, ok = ctx.nid_finish(bins)
        if self.world > 1:
            # every rank derives (cost, gradient) from the same summed histograms, so every rank's optimiser takes the
            # same steps without a broadcast -- IF the all-reduce hands every rank the same bits.  Checked instead of
            # assumed: a rank that saw different numbers would walk a different line search and the ranks would wait
            # for each other in different collectives for ever.
            import torch
            import torch.distributed as dist

            mine = torch.tensor([cost, *[float(g) for g in grad], float(ok)], dtype=torch.float64).view(torch.int64)  # bit patterns: NaN == NaN
            lo, hi = mine.clone(), mine.clone()
            if dist.get_backend(self.group) == "nccl":
                lo, hi = lo.cuda(), hi.cuda()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo.cpu(), hi.cpu()):
                raise RuntimeError("NID cost / gradient differ between the ranks after the all-reduce of the histograms: "
                                   f"min {lo.tolist()} max {hi.tolist()}")
        return cost, grad, ok

    def calibrate(self, T_init=None, bins: int = 16, max_outer_iterations: int = 10):
        """T_camera_lidar_optimized (4x4), final cost, evaluations -- identity initial guess as calibrate.cpp:45-51"""
        Ti = np.eye(4) if T_init is None else np.asarray(T_init, np.float64)
        if self.world == 1:
            return self.engine.ctx.nid_optimize(Ti, bins, max_outer_iterations)
        return self.engine.ctx.nid_optimize_with(self.evaluate, Ti, bins, max_outer_iterations)


class CloudSmooth:
    """CloudSmooth::process: MovingLeastSquares (+ optional SOR brackets)."""

    def __init__(self, engine: HipEngine, params: capi.MLSParams | None = None):
        self.engine = engine
        self.params = params if params is not None else capi.default_mls_params()

    def process_sharded(self, n_total: int, rank: int, world: int, group=None):
        """MLS (upsampling NONE) with the queries dealt out over `world` ranks by slabs of the stage's own spatial order
        (whole wavefronts: 1 / world of the work whatever order the caller's points come in): every rank holds the whole
        cloud, fits its slab, the variable-length results are all-gathered and merged by source index.
        Returns the full result, in input order, on every rank."""
        ctx = self.engine.ctx
        local = ctx.mls_fetch(ctx.mls_process_slab(self.params, rank, world))
        if world == 1:
            return local
        import torch
        import torch.distributed as dist

        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        m = torch.tensor([len(local["index"])], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(m) for _ in range(world)]
        dist.all_gather(counts, m, group=group)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        packed = np.zeros((cap, 8), np.float32)  # xyz(3) normal(3) curvature index(as int bits)
        k = len(local["index"])
        packed[:k, 0:3] = local["xyz"]
        packed[:k, 3:6] = local["normal"]
        packed[:k, 6] = local["curvature"]
        packed[:k, 7] = local["index"].view(np.float32)
        mine = torch.from_numpy(packed).to(dev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine, group=group)
        parts = [outs[r].cpu().numpy()[: counts[r]] for r in range(world)]
        allp = np.concatenate(parts, axis=0) if parts else np.zeros((0, 8), np.float32)
        idx = np.ascontiguousarray(allp[:, 7]).view(np.int32)
        allp = allp[np.argsort(idx, kind="stable")]  # a point is fitted by exactly one slab: the merge restores input order
        return dict(xyz=np.ascontiguousarray(allp[:, 0:3]), normal=np.ascontiguousarray(allp[:, 3:6]),
                    curvature=np.ascontiguousarray(allp[:, 6]), index=np.ascontiguousarray(allp[:, 7]).view(np.int32))

    def outlier_removal_sharded(self, n_total: int, rank: int, world: int, group=None):
        """StatisticalOutlierRemoval (cloudSmooth.cpp:109-116) with the queries dealt out by slabs of the filter's own spatial
        order: every rank holds the whole cloud, computes the mean distances and per-chunk (sum, sum of squares) of its
        slab, the chunk sums are put together (ceil(n / 16384) pairs of doubles), every rank classifies its slab and the
        keep flags are combined.  Equal to ctx.sor() on one GPU bit for bit.  Returns the full keep mask."""
        ctx = self.engine.ctx
        c = ctx.sor_chunk_points()
        chunks = (n_total + c - 1) // c
        sums = np.zeros((chunks, 2), np.float64)
        first, mine = ctx.sor_partial(self.params.sor_mean_k, rank, world)
        sums[first:first + len(mine)] = mine
        if world > 1:
            import torch
            import torch.distributed as dist

            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            # every chunk is owned by exactly one rank and zero elsewhere: x + 0.0 is exact, the sum IS the concatenation
            t = torch.from_numpy(sums).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            sums = t.cpu().numpy()
        keep = ctx.sor_finish(self.params.sor_std_mul, sums, rank, world)[0]
        if world > 1:
            t = torch.from_numpy(keep).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            keep = t.cpu().numpy()
        return keep

    def process(self, with_outlier_removal: bool = True):
        """SOR -> MLS (+ upsampling) -> SOR as cloudSmooth.cpp:109-164; MLS alone when asked.  An upsampled cloud that one
        result cannot hold (the reference's VOXEL_GRID_DILATION 1 mm x 4 on a real map) goes through the streamed form and is
        gathered on the host, as the C++ shim does (host/pcp_shim.hpp)."""
        ctx = self.engine.ctx
        if not with_outlier_removal:
            return ctx.mls_fetch(ctx.mls_process(self.params))
        try:
            return ctx.mls_fetch(ctx.cloud_smooth(self.params))
        except capi.PcpError as e:
            if e.code != capi.PCP_ERR_NOMEM or self.params.upsampling != 3:
                raise
        parts = list(self.process_streamed(1 << 28))
        if not parts:
            return ctx.mls_fetch(0)
        import numpy as np

        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def process_streamed(self, chunk_capacity: int = 1 << 28, hold_device_memory: bool = False):
        """The whole CloudSmooth::process through pcp_cloud_smooth_stream_*: yields the survivors of the trailing outlier removal
        chunk by chunk (dicts as mls_fetch returns them), in the order the one-shot form returns them; `self.streamed` holds
        (rows before the last filter, rows kept, chunks) and the stream's diagnostics."""
        ctx = self.engine.ctx
        total, kept, chunks = ctx.cloud_smooth_stream_begin(self.params, chunk_capacity)
        self.streamed = {"rows": total, "kept": kept, "chunks": chunks, **ctx.cloud_smooth_stream_stats()}
        try:
            while True:
                m = ctx.cloud_smooth_stream_next()
                if m == 0:
                    break
                yield ctx.mls_fetch(m)
        finally:
            if not hold_device_memory:
                ctx.cloud_smooth_stream_end()
