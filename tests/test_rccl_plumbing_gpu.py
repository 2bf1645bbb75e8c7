"""The torch.distributed plumbing of the multi-GPU path on one GPU: the depth maps the
library owns are wrapped (no copy) as a torch tensor and all-reduced with RCCL
(backend "nccl", world_size 1 here; the N > 1 arithmetic is covered by the gloo test)."""
import os
import socket

import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu


def test_depth_maps_alias_and_rccl_allreduce(oracle, small_scene):
    import torch
    import torch.distributed as dist

    from pointcloudprocessor_amd import capi, pipeline

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        eng = pipeline.HipEngine(0)
        eng.configure(small_scene["cam"])
        eng.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
        eng.set_keyframes(small_scene["poses"], small_scene["images"])
        col = pipeline.PointCloudColorizer(eng, 0, 1)
        eng.depth_pass()
        t = eng.depth_maps_tensor()
        mh, mw = eng.ctx.map_shape
        assert t.is_cuda and t.dtype == torch.float32 and t.numel() == len(small_scene["poses"]) * mh * mw
        before = eng.ctx.download_depth_map(2)
        assert np.array_equal(t.view(len(small_scene["poses"]), mh, mw)[2].cpu().numpy(), before)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        assert np.array_equal(eng.ctx.download_depth_map(2), before)
        # the tensor aliases the library's buffer: a write through torch is seen by the library
        t[2 * mh * mw] = 0.125
        torch.cuda.synchronize()
        assert eng.ctx.download_depth_map(2)[0, 0] == np.float32(0.125)
        # full sharded driver at world 1 == plain colorize
        a = col.run()
        b = eng.ctx.colorize()
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["has"], b["has"])
        eng.close()
    finally:
        dist.destroy_process_group()


def test_bench_two_rank_path_runs_to_completion():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, one rank per process), with the
    gloo backend so that both ranks can share this box's single GPU: every collective of the step, of the timing leg
    and of the shutdown must be entered by all ranks (a rank-0-only step once deadlocked here)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--points", "200000",
           "--frames", "8", "--steps", "1", "--warmup", "1", "--roofline-points", "2000000", "--settle-ms", "0", "--verify",
           "--cpu-points", "100000", "--cpu-frames", "2", "--sharded-legs-points", "150000", "--sharded-legs-frames", "5"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=root)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["config"]["points_total"] == 400000 and "configs[3]" in line["config"]["workload"]
    assert line["roofline"]["frac"] > 0  # the roofline leg stays on rank 0 at N > 1
    assert "rehearsal" in line
    # --verify: the two shards' colours, all-gathered, equal a one-GPU run of the whole 400 k-point map on rank 0
    assert line["verify"]["equal_to_one_gpu_run"] is True and line["verify"]["coloured"] > 0
    assert line["cpu_baseline"]["value"] > 0  # reported at N > 1 as well (rank 0)
    # the hidden_points_removal and smoothing legs of N > 1 (what the first multi-GPU lease times): ran on both ranks
    legs = line["sharded_legs"]
    assert legs["hpr"]["hpr_ms"] > 0 and legs["hpr"]["hull_vertices"] > 0 and legs["hpr"]["keyframe0_shards_equal_owner"] is True
    assert legs["smooth"]["smooth_ms"] > 0 and legs["smooth"]["kept"] > 0 and legs["smooth"]["same_mask_on_every_rank"] is True
    # the verdicts travel as bit-packed slices: three rounds (5 keyframes over 2 ranks) of at most n / 8 bytes (+ padding) per rank
    assert 0 < legs["hpr"]["exchange_bytes_per_rank"] <= 3 * (150000 // 8 + 2)
