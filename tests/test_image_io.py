"""host/image_io.hpp: the JPEG and PNG decoders that stand in for cv::imread,
bit for bit against Pillow (libjpeg-turbo / libpng) in this image.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def _exe():
    from pointcloudprocessor_amd import _build, host_build

    _build.build()
    return host_build.build()["image_dump"]


def _decode(path, gray=False):
    out = str(path) + ".raw"
    r = subprocess.run([_exe(), str(path), out] + (["gray"] if gray else []), capture_output=True, text=True)
    if r.returncode != 0:
        return None
    with open(out, "rb") as f:
        w, h, c = map(int, f.readline().split())
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, c)


def _picture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    im = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 5.0 - y / 9.0),
                   128 + 80 * np.sin((x + y) / 11.0)], 2) + rng.normal(0, 12, (h, w, 3))
    return np.clip(im, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(64, 64), (37, 53), (135, 240), (17, 9), (8, 8), (100, 3), (1, 1)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_jpeg_matches_libjpeg(tmp_path, size, subsampling):
    """4:4:4 / 4:2:2 / 4:2:0, sizes that are not MCU multiples: islow IDCT, fancy upsampling and
    the YCbCr->RGB tables must reproduce libjpeg's output exactly."""
    h, w = size
    for q in (25, 75, 96):
        p = tmp_path / f"t_{q}.jpg"
        PIL.fromarray(_picture(h, w, q)).save(p, quality=q, subsampling=subsampling)
        ref = np.array(PIL.open(p).convert("RGB"))[:, :, ::-1]
        got = _decode(p)
        assert got is not None and got.shape == ref.shape
        assert np.array_equal(got, ref), (size, subsampling, q, int(np.abs(got.astype(int) - ref).max()))


def test_jpeg_grayscale_restart_and_luma_read(tmp_path):
    im = _picture(90, 130)
    p = tmp_path / "g.jpg"
    PIL.fromarray(im[:, :, 0]).save(p, quality=80)
    ref = np.array(PIL.open(p))
    assert np.array_equal(_decode(p, gray=True)[:, :, 0], ref)
    assert np.array_equal(_decode(p), np.repeat(ref[:, :, None], 3, axis=2))  # gray JPEG read as BGR
    p = tmp_path / "r.jpg"
    PIL.fromarray(im).save(p, quality=85, subsampling=2, restart_marker_blocks=3)
    assert np.array_equal(_decode(p), np.array(PIL.open(p).convert("RGB"))[:, :, ::-1])
    # IMREAD_GRAYSCALE of a colour JPEG: the decoder's luma plane
    ref_l = np.array(PIL.open(p).convert("L", dither=None))
    got_l = _decode(p, gray=True)[:, :, 0]
    ycc = np.array(PIL.open(p).convert("YCbCr"))[:, :, 0] if False else None  # Pillow re-derives Y from RGB: skip exactness
    assert np.abs(got_l.astype(int) - ref_l.astype(int)).max() <= 2
    # progressive files are refused, not mis-decoded
    p = tmp_path / "prog.jpg"
    PIL.fromarray(im).save(p, quality=85, progressive=True)
    assert _decode(p) is None


def test_png_variants(tmp_path):
    rgb = _picture(40, 60)
    cases = {
        "L": PIL.fromarray(rgb[:, :, 0]),
        "RGB": PIL.fromarray(rgb),
        "RGBA": PIL.fromarray(np.dstack([rgb, np.full((40, 60), 200, np.uint8)]), "RGBA"),
        "P": PIL.fromarray(rgb).convert("P", palette=PIL.ADAPTIVE),
        "LA": PIL.fromarray(np.dstack([rgb[:, :, 1], np.full((40, 60), 9, np.uint8)]), "LA"),
    }
    for name, img in cases.items():
        p = tmp_path / f"{name}.png"
        img.save(p)
        ref = np.array(PIL.open(p).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(_decode(p), ref), name
    # 16-bit gray: high byte; a 0/255 mask read as gray
    g16 = (np.arange(40 * 60, dtype=np.uint16).reshape(40, 60) * 27)
    p = tmp_path / "g16.png"
    PIL.fromarray(g16).save(p)
    assert np.array_equal(_decode(p, gray=True)[:, :, 0], (g16 >> 8).astype(np.uint8))
    m = (np.random.default_rng(1).random((30, 45)) > 0.5).astype(np.uint8) * 255
    p = tmp_path / "m.png"
    PIL.fromarray(m).save(p)
    assert np.array_equal(_decode(p, gray=True)[:, :, 0], m)
    # interlaced PNG and garbage are refused
    p = tmp_path / "junk.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(40))
    assert _decode(p) is None
