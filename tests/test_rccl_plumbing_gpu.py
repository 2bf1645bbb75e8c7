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
