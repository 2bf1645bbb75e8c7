import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Factory of pcp contexts on cuda:0; fails loudly when the HIP extension is missing."""
    from pointcloudprocessor_amd import capi

    made = []

    def make():
        ctx = capi.Context(0)
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_capi

    oracle_capi.build()
    return oracle_capi


def cam_struct(module, d: dict):
    cam = module.Camera()
    for k, _ in module.Camera._fields_:
        setattr(cam, k, d[k])
    return cam


@pytest.fixture(scope="session")
def small_scene():
    """20k points, 6 keyframes, 480x270 camera: finishes in well under a second on the oracle."""
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict("tiny")
    x, y, z, _ = synth.make_cloud(20000)
    poses, ts = synth.make_trajectory(6)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(6)]
    masks = [synth.make_mask(f, cd["image_width"], cd["image_height"]) for f in range(6)]
    return dict(cam=cd, x=x, y=y, z=z, poses=poses, ts=ts, images=imgs, masks=masks)
