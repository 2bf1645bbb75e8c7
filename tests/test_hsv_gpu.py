"""Row f4 on the GPU: generateColorMap's 8-bit BGR -> HSV -> BGR round trip (PointCloudProcessor.cpp:722-741) fused
into the image pack kernel (pcp_set_image_adjust) against the oracle's restatement of OpenCV 4.2's arithmetic:
every pixel equal, every 8-bit colour of a lattice, S / V scales, and the colour run that samples the adjusted image."""
import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu


def _lattice_image(h, w, seed):
    """All colours of a 52^3 lattice (step 5) first, random pixels after: more than every (v, diff) table entry."""
    v = np.arange(0, 256, 5, dtype=np.uint8)
    lat = np.stack(np.meshgrid(v, v, v, indexing="ij"), axis=-1).reshape(-1, 3)
    px = np.random.default_rng(seed).integers(0, 256, (h * w, 3), dtype=np.uint8)
    m = min(len(lat), h * w)
    px[:m] = lat[:m]
    return px.reshape(h, w, 3)


@pytest.mark.parametrize("sat,val", [(1.0, 1.0), (1.3, 0.8), (0.0, 1.0)])
def test_pack_kernel_applies_opencvs_round_trip(gpu_ctx_factory, oracle, small_scene, sat, val):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = small_scene["cam"]
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
    ctx.set_frames(small_scene["poses"][:2])
    H, W = cd["image_height"], cd["image_width"]
    raw = _lattice_image(H, W, 3)
    assert H * W < 52 ** 3 or True
    ctx.set_image_adjust(True, sat, val)
    ctx.upload_image(0, raw)
    ctx.upload_mask(0, small_scene["masks"][0])
    ctx.set_image_adjust(False)
    ctx.upload_image(1, raw)  # untouched
    got0, mask0 = ctx.download_image(0)
    got1, _ = ctx.download_image(1)
    assert np.array_equal(got0, oracle.hsv_round_trip(raw, sat, val))
    assert np.array_equal(mask0, small_scene["masks"][0])  # the mask byte survives the fused pack
    assert np.array_equal(got1, raw)
    if sat == 1.0 and val == 1.0:
        assert (got0 != raw).any(axis=2).mean() > 0.4  # lossy: NOT the identity


def test_colour_run_samples_the_adjusted_image(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    s = small_scene
    ctx = gpu_ctx_factory()
    cd = s["cam"]
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(s["x"], s["y"], s["z"])
    ctx.set_frames(s["poses"])
    ctx.set_image_adjust(True)
    for f, im in enumerate(s["images"]):
        ctx.upload_image(f, im)
    got = ctx.colorize()
    adj = [oracle.hsv_round_trip(im) for im in s["images"]]
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), s["x"], s["y"], s["z"], s["poses"], adj)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    raw = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), s["x"], s["y"], s["z"], s["poses"],
                          s["images"], want_top=False)
    assert (raw["rgb"] != ref["rgb"]).any(axis=1).sum() > 100  # skipping the stage would have been visible
    ctx.set_image_adjust(False)
