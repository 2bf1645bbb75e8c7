"""The kernels' short exact division sequences against the compiler's IEEE `/` (pcp_selftest_arithmetic).

pinhole.hpp:17-18 divides x/z and y/z in fp64, view_culling.cpp:88 divides the projected pixel by 14 in fp32;
pcp_device.hpp computes the same correctly rounded quotients with fewer instructions.  The self-test runs both
forms on the device: random float triples over every exponent for the fp64 pair, all 2^32 fp32 bit patterns for
the constant division.  Any disagreement would make pixel / cell indices differ from the reference's.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ds", [14, 1, 3, 7, 10, 16, 100])
def test_short_divisions_equal_ieee_division(gpu_ctx_factory, ds):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cull = capi.default_cull_params()
    cull.downsample_factor = ds
    ctx.set_camera(capi.default_camera(), cull)
    bad64, bad32 = ctx.selftest_arithmetic(samples=1 << 27, seed=20241008 + ds)
    assert bad64 == 0
    assert bad32 == 0
    ctx.close()


def test_fast_and_plain_arithmetic_give_identical_results(gpu_ctx_factory, small_scene, monkeypatch):
    """End to end: PCP_DISABLE_FAST_EXACT=1 (plain `/` everywhere) and the default build agree bit for bit."""
    from pointcloudprocessor_amd import capi
    from conftest import cam_struct

    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("PCP_DISABLE_FAST_EXACT", flag)
        ctx = gpu_ctx_factory()
        ctx.set_camera(cam_struct(capi, small_scene["cam"]))
        ctx.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
        ctx.set_frames(small_scene["poses"])
        for f, im in enumerate(small_scene["images"]):
            ctx.upload_image(f, im)
        res = ctx.colorize()
        proj = [ctx.project_frame(f) for f in range(len(small_scene["poses"]))]
        outs.append((res, proj))
        ctx.close()
    (r0, p0), (r1, p1) = outs
    assert np.array_equal(r0["rgb"], r1["rgb"]) and np.array_equal(r0["has"], r1["has"])
    for a, b in zip(p0, p1):
        assert np.array_equal(a["cell"], b["cell"]) and np.array_equal(a["pixel"], b["pixel"])
