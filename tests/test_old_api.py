"""the reference's historic entry points over `long` samples (old/parallel_*.c) on the GPU path
(csrc/old_api.hip, gh_old_parallel_*).  Parity status UNPINNED: old/*.c include a utils.h the reference does not contain,
so they cannot be compiled here, and the reference holds no fixture for them.  Checked against (a) the oracle's restatement of
the three functions on long arrays, (b) the answers SURVEY.md 8(c) recorded when it compiled them with a stand-in header
(tests/golden/known_answers.json: old_kernels)."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN  # noqa: F401

GAUSS = [1, 2, 1, 2, 4, 2, 1, 2, 1]


def test_oracle_restatement_matches_the_survey_answers(oracle, known):
    """CPU only: the long-array restatement reproduces the recorded answers on rand(1) frames."""
    for name, k in known["old_kernels"].items():
        if not isinstance(k, dict):
            continue
        w, h = k["w"], k["h"]
        u8 = oracle.frame_rand(w, h, 3, seed=1)
        img = u8.astype(np.int64)
        assert oracle.old_parallel_avg_pixel(img) == k["avg"]
        g, mm = oracle.old_parallel_to_grayscale(img)
        assert mm == k["mean_gray_minmax"]
        assert hashlib.sha256(g.astype(np.uint8).tobytes()).hexdigest() == k["mean_gray_rgb"]
        pad = np.zeros((h + 2, w + 2, 3), dtype=np.int64)
        pad[1:-1, 1:-1] = img
        out = oracle.old_parallel_convolution(pad, GAUSS, np.zeros((h, w, 3), dtype=np.int64), 16)
        assert hashlib.sha256(out.astype(np.uint8).tobytes()).hexdigest() == k["conv_gauss16"]
        out = oracle.old_parallel_convolution(pad, [1] * 9, np.zeros((h, w, 3), dtype=np.int64), 9)
        assert hashlib.sha256(out.astype(np.uint8).tobytes()).hexdigest() == k["conv_box9"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rand1_1024x1024", "rand1_1021x517"])
def test_historic_entry_points_reproduce_the_survey_answers(pkg, oracle, known, name):
    """Byte-valued frames: narrowed to u8 and run on the hot-path kernels (conv3x3 strip kernel, __shfl reduction)."""
    k = known["old_kernels"][name]
    w, h = k["w"], k["h"]
    img = oracle.frame_rand(w, h, 3, seed=1).astype(np.int64)
    api = pkg.hostapi
    assert api.old_parallel_avg_pixel(img) == k["avg"]
    g, mm = api.old_parallel_to_grayscale(img)
    assert mm == k["mean_gray_minmax"]
    assert hashlib.sha256(g.astype(np.uint8).tobytes()).hexdigest() == k["mean_gray_rgb"]
    pad = np.zeros((h + 2, w + 2, 3), dtype=np.int64)
    pad[1:-1, 1:-1] = img
    out = api.old_parallel_convolution(pad, GAUSS, np.zeros((h, w, 3), dtype=np.int64), 16)
    assert hashlib.sha256(out.astype(np.uint8).tobytes()).hexdigest() == k["conv_gauss16"]
    out = api.old_parallel_convolution(pad, [1] * 9, np.zeros((h, w, 3), dtype=np.int64), 9)
    assert hashlib.sha256(out.astype(np.uint8).tobytes()).hexdigest() == k["conv_box9"]


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (37, 53), (64, 1040), (130, 257)])
def test_historic_semantics_beyond_bytes(pkg, oracle, rows, cols):
    """What the u8 kernels cannot express goes through the 64-bit kernels: non-zero padding, negative taps, a non-zero output
    to accumulate into, samples outside 0..255, accumulate-into avgs -- always equal to the restated reference arithmetic."""
    rng = np.random.RandomState(rows * 1000 + cols)
    api = pkg.hostapi
    for case in range(4):
        lo, hi = [(0, 256), (0, 256), (-500, 70000), (-3, 4)][case]
        pad = rng.randint(lo, hi, size=(rows + 2, cols + 2, 3)).astype(np.int64)          # padding is NOT zero
        taps = [GAUSS, [-1, 0, 1, -2, 0, 2, -1, 0, 1], [3, -7, 2, 0, 11, -4, 5, 1, -9], [0] * 8 + [1]][case]
        norm = [16, 1, -3, 7][case]
        prev = rng.randint(-50, 50, size=(rows, cols, 3)).astype(np.int64) if case else np.zeros((rows, cols, 3), dtype=np.int64)
        want = oracle.old_parallel_convolution(pad, taps, prev.copy(), norm)
        got = api.old_parallel_convolution(pad, taps, prev.copy(), norm)
        assert np.array_equal(got, want), ("conv", case)
        img = np.ascontiguousarray(pad[1:-1, 1:-1])
        avgs_in = [0, 0, 0] if case == 0 else [int(v) for v in rng.randint(-1000, 1000, size=3)]
        assert api.old_parallel_avg_pixel(img, avgs_in) == oracle.old_parallel_avg_pixel(img, avgs_in), ("avg", case)
        g, mm = api.old_parallel_to_grayscale(img)
        wg, wmm = oracle.old_parallel_to_grayscale(img)
        assert mm == wmm and np.array_equal(g, wg), ("gray", case)


@pytest.mark.gpu
def test_historic_entry_points_degenerate_and_errors(pkg):
    api = pkg.hostapi
    out, mm = api.old_parallel_to_grayscale(np.zeros((0, 5, 3), dtype=np.int64))
    assert mm == [256, -1]                                    # old/parallel_to_grayscale.c:9-10 on an empty frame
    with pytest.raises(pkg.GrayHipError):
        api.old_parallel_convolution(np.zeros((3, 3, 3), dtype=np.int64), GAUSS, np.zeros((1, 1, 3), dtype=np.int64), 0)
    with pytest.raises(pkg.GrayHipError):
        api.old_parallel_avg_pixel(np.full((2, 2, 3), 1 << 40, dtype=np.int64))
