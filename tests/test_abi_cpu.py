"""CPU suite: libpcp_hip.so builds (hipcc cross-compiles gfx950), loads, exports
every symbol include/pcp_hip.h declares, fails loudly without a GPU, and its
host-only helpers agree with the oracle.  No compute calls."""
import ctypes as C

import numpy as np
import pytest


def test_library_builds_loads_and_exports_every_declared_symbol():
    from pointcloudprocessor_amd import _build, capi

    path = _build.build()
    lib = capi.load()
    names = capi.declared_symbols()
    assert len(names) >= 30 and "pcp_colorize" in names and "pcp_mls_process" in names
    missing = [s for s in names if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.pcp_abi_version() == 6
    assert path.endswith("libpcp_hip.so")


def test_object_targets_gfx950_only():
    import re

    from pointcloudprocessor_amd import _build

    blob = open(_build.build(), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_create_fails_loudly_without_gpu():
    import torch

    from pointcloudprocessor_amd import capi

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.PcpError) as e:
        capi.Context(0)
    assert e.value.code == capi.PCP_ERR_DEVICE and "no CPU fallback" in str(e.value)


def test_defaults_match_reference_constants(oracle):
    from pointcloudprocessor_amd import capi

    a, b = capi.default_camera(), oracle.default_camera()
    for k, _ in capi.Camera._fields_:
        assert getattr(a, k) == getattr(b, k), k
    a, b = capi.default_cull_params(), oracle.default_cull_params()
    for k, _ in capi.CullParams._fields_:
        assert getattr(a, k) == getattr(b, k), k
    m = capi.default_mls_params()
    assert (m.search_radius, m.sqr_gauss_param, m.polynomial_order, m.compute_normals, m.upsampling) == (
        0.03, 0.0009, 2, 1, 3)
    assert (m.vgd_iterations, m.sor_mean_k, m.sor_std_mul) == (4, 60, 0.7)
    assert abs(m.vgd_voxel_size - 0.001) < 1e-9


def test_pose_to_matrices_host_helper_matches_oracle(oracle, small_scene):
    from pointcloudprocessor_amd import capi

    rng = np.random.default_rng(1)
    for pose in small_scene["poses"]:
        a, b = capi.pose_to_matrices(pose), oracle.pose_to_matrices(pose)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        pose = np.concatenate([rng.uniform(-50, 50, 3), q])
        T = np.eye(4)
        T[:3, :3] += rng.normal(0, 0.01, (3, 3))
        T[:3, 3] = rng.normal(0, 0.05, 3)
        for t in (None, T):
            a, b = capi.pose_to_matrices(pose, t), oracle.pose_to_matrices(pose, t)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_kernel_names():
    from pointcloudprocessor_amd import capi

    lib = capi.load()
    names = [lib.pcp_kernel_name(C.c_int32(k)).decode() for k in range(capi.K_COUNT)]
    assert names[0] == "project_frame" and len(set(names)) == capi.K_COUNT


def test_every_entry_point_is_documented():
    """INTEGRATION.md names every symbol include/pcp_hip.h declares (with the reference call site it replaces)."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "pcp_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(pcp_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 40
    missing = [n for n in names if n not in doc and not n.startswith(("pcp_timing_", "pcp_default_"))]
    assert not missing, missing
    assert "pcp_timing_" in doc and "pcp_default_" in doc


def test_public_header_is_plain_c(tmp_path):
    """include/pcp_hip.h is the FFI surface: it must compile as C99 (cgo / ctypes / JNI generators) and as C++11."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "pcp_hip.h"\nint main(void) { return pcp_abi_version() > 0 ? 0 : 1; }\n')
    inc = os.path.join(root, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o",
                    str(tmp_path / "abi_c.o")], check=True, capture_output=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o",
                    str(tmp_path / "abi_cpp.o")], check=True, capture_output=True)
