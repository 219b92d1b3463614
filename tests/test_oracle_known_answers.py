"""The oracle (oracle/pixel_oracle.c) against the known answers of the COMPILED REFERENCE
(tests/golden/known_answers.json) -- this is what pins the oracle."""
import numpy as np
import pytest

from conftest import make_frame

SMALL = ["cube4096", "smooth4096", "rand1_4096", "smooth1021x517"]
FULL = ["rand1_16384", "smooth16384"]


@pytest.mark.parametrize("name", SMALL + FULL)
def test_generators_and_hot_path_hashes(oracle, known, name):
    spec = known["frames"][name]
    frame = make_frame(oracle, name, spec)
    assert oracle.sha256(frame) == spec["in"]
    for mode_name, mode in (("IEEE", oracle.MODE_IEEE), ("FMA", oracle.MODE_FMA)):
        want = known[mode_name][name]
        gray = oracle.gray_to_plane(frame, mode)
        assert oracle.sha256(gray) == want["gray"], (mode_name, "gray")
        if name in SMALL:   # the in-place form triples the bytes hashed; the plane pins the same arithmetic
            assert oracle.sha256(oracle.gray_inplace(frame, mode)) == want["grayrgb"], (mode_name, "grayrgb")
        edge = oracle.sobel(gray)
        assert oracle.sha256(edge) == want["edge"], (mode_name, "edge")
        if "edge_sum" in want:
            assert int(edge.sum(dtype=np.uint64)) == want["edge_sum"]
        del gray, edge
    if name in known["channel_sums"]:
        assert oracle.sum_rgb(frame) == known["channel_sums"][name]


def test_rgba_alpha_untouched(oracle, known):
    ka = known["rgba_rand1_1024"]
    img = oracle.frame_rand(1024, 1024, 4, seed=1)
    assert oracle.sha256(img) == ka["in"]
    for mode_name, mode in (("IEEE", oracle.MODE_IEEE), ("FMA", oracle.MODE_FMA)):
        out = oracle.gray_inplace(img, mode)
        assert oracle.sha256(out) == ka[mode_name]
        assert np.array_equal(out[..., 3], img[..., 3])


@pytest.mark.parametrize("name", ["rand1_1024x1024", "rand1_1021x517"])
def test_old_kernels_restated_on_u8(oracle, known, name):
    ka = known["old_kernels"][name]
    w, h = ka["w"], ka["h"]
    img = oracle.frame_rand(w, h, 3, seed=1)
    assert oracle.sha256(img) == ka["in"]
    sums = oracle.sum_rgb(img)
    assert sums == ka["sums"]
    assert oracle.avg_from_sums(sums, w * h) == ka["avg"]
    out, lo, hi = oracle.mean_gray_minmax(img)
    assert [lo, hi] == ka["mean_gray_minmax"]
    assert oracle.sha256(out) == ka["mean_gray_rgb"]
    assert oracle.sha256(oracle.conv3x3(img, [1, 2, 1, 2, 4, 2, 1, 2, 1], 16)) == ka["conv_gauss16"]
    assert oracle.sha256(oracle.conv3x3(img, [1] * 9, 9)) == ka["conv_box9"]


def test_gray_of_gray_fixed_points(oracle, known):
    v = np.arange(256, dtype=np.uint8)
    img = np.repeat(v[:, None], 3, axis=1).reshape(1, 256, 3)
    out = oracle.gray_to_plane(img, oracle.MODE_IEEE)[0]
    moved = [int(i) for i in np.nonzero(out != v)[0]]
    assert moved == known["gray_of_gray_not_fixed_ieee"]
    assert all(out[i] == i - 1 for i in moved)


def test_float_sqrt_equals_isqrt_for_every_reachable_m(oracle):
    # sobel.c:20  (int)sqrtf((float)m) for m <= 2*1020^2; numpy's float32 sqrt is correctly rounded like libm's
    m = np.arange(0, 2 * 1020 * 1020 + 1, dtype=np.int64)
    f = np.sqrt(m.astype(np.float32)).astype(np.int64)
    isq = np.floor(np.sqrt(m.astype(np.float64))).astype(np.int64)
    isq = np.where(isq * isq > m, isq - 1, isq)
    isq = np.where((isq + 1) * (isq + 1) <= m, isq + 1, isq)
    assert np.array_equal(f, isq)
    for probe in (0, 1, 2, 3, 4, 65024, 65025, 65026, 2080800):
        assert oracle.lib().orc_isqrt(probe) == isq[probe]


def test_sobel_degenerate_sizes(oracle):
    for w, h in ((1, 1), (2, 5), (5, 2), (3, 3), (1, 7)):
        plane = oracle.frame_rand(w, h, 1, seed=3)[..., 0]
        out = oracle.sobel(plane)
        if w < 3 or h < 3:
            assert not out.any()
        else:
            assert out[0].sum() == 0 and out[-1].sum() == 0 and out[:, 0].sum() == 0 and out[:, -1].sum() == 0
    # interior-only form leaves the border bytes of dst alone
    plane = oracle.frame_rand(9, 7, 1, seed=4)[..., 0]
    dst = np.full((7, 9), 0xAB, dtype=np.uint8)
    oracle.sobel(plane, dst=dst)
    assert (dst[0] == 0xAB).all() and (dst[-1] == 0xAB).all() and (dst[:, 0] == 0xAB).all() and (dst[:, -1] == 0xAB).all()
    assert np.array_equal(dst[1:-1, 1:-1], oracle.sobel(plane)[1:-1, 1:-1])
