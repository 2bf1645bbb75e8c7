"""Row f4: the 8-bit BGR -> HSV -> BGR round trip generateColorMap applies to every keyframe
(PointCloudProcessor.cpp:722-741), OpenCV 4.2 arithmetic restated in oracle/ (C and numpy twins).  Known answers:
greys, primaries, saturation 0 / 255, hue wrap; twin agreement on every colour of a lattice.  No OpenCV in this image,
so parity stays unpinned; these tests pin the restatement against itself and against hand calculation."""
import numpy as np


def test_twins_agree_on_a_colour_lattice_and_random_pixels(oracle):
    from oracle import np_oracle as npo

    v = np.arange(0, 256, 5, dtype=np.uint8)
    lat = np.stack(np.meshgrid(v, v, v, indexing="ij"), axis=-1).reshape(-1, 3)
    rnd = np.random.default_rng(7).integers(0, 256, (300_000, 3), dtype=np.uint8)
    for px in (lat, rnd):
        a, b = oracle.hsv_round_trip(px), npo.hsv_round_trip(px)
        assert np.array_equal(a, b)
        assert np.abs(a.astype(int) - px.astype(int)).max() <= 6  # lossy, but close: 8-bit HSV quantisation
    assert (oracle.hsv_round_trip(rnd) != rnd).any(axis=1).mean() > 0.5  # NOT the identity


def test_known_answers(oracle):
    rt = oracle.hsv_round_trip
    greys = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, axis=1)
    assert np.array_equal(rt(greys), greys)  # s == 0: b = g = r = v, and v/255*255 rounds back for every level
    prim = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255]], np.uint8)
    assert np.array_equal(rt(prim), prim)  # saturation 255, hues 120 / 60 / 0 / 90 / 150 / 30: sector boundaries
    # hue wrap: red with a little more blue than green has h = (g - b) * 30 / diff < 0 -> + 180
    # b, g, r = 10, 0, 200: v = 200, diff = 200, s = 255, h = round(-10 * 30 / 200) = round(-1.5 + eps) ...
    one = rt(np.array([[10, 0, 200]], np.uint8))[0]
    # by hand: hdiv[200] = rint(180*4096/1200) = 614; h = (-10*614 + 2048) >> 12 = floor(-0.999) = -1 -> 179
    # back: h = 179/30 = 5.9667, sector 5, f = .9667: (b, g, r) = (tab2, tab1, tab0) = (v(1 - s f), v(1 - s), v)
    # s = 255/255 = 1, v = 200/255: b = 200 * (1 - .96667) = 6.67 -> 7, g = 0, r = 200
    assert one.tolist() == [7, 0, 200]
    # scales: brightness 0 blacks the image out, saturation 0 makes it grey at V = max(b, g, r)
    px = np.array([[12, 200, 90], [255, 1, 77]], np.uint8)
    assert np.array_equal(rt(px, 1.0, 0.0), np.zeros_like(px))
    assert np.array_equal(rt(px, 0.0, 1.0), np.array([[200] * 3, [255] * 3], np.uint8))
    # saturate_cast: V * 1.5 clamps at 255 (rounds half to even below it)
    assert rt(np.array([[100, 100, 100]], np.uint8), 1.0, 1.5).tolist() == [[150, 150, 150]]
    assert rt(np.array([[200, 200, 200]], np.uint8), 1.0, 1.5).tolist() == [[255, 255, 255]]


def test_forward_tables_are_opencvs(oracle):
    """sdiv_table[i] = cvRound((255 << 12) / (1. * i)), hdiv_table180[i] = cvRound((180 << 12) / (6. * i)): a few
    entries by hand, and the property that makes S exact for fully saturated colours."""
    assert int(np.rint((255 << 12) / 255.0)) == 4096 and int(np.rint((180 << 12) / (6.0 * 255))) == 482
    sat = np.array([[0, 0, k] for k in range(1, 256)], np.uint8)  # pure reds of every brightness: S = 255, H = 0
    out = oracle.hsv_round_trip(sat)
    assert np.array_equal(out, sat)
