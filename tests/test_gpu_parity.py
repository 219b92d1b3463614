"""Parity tests proper: the gfx950 kernels, called through the C ABI (libgrayhip.so), against the
oracle, the compiled reference's known answers and the reference's golden image.  Bit-exact for
every uint8 output and every integer sum -- there is no floating-point tolerance anywhere."""
import hashlib
import os

import numpy as np
import pytest

from conftest import LOOPBACK_MODES, loopback_pad, loopback_sobel

from conftest import make_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod(pkg):
    import torch
    pkg.hostapi.init(0)
    return torch


def dev(torch, a: np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sha(t) -> str:
    return hashlib.sha256(t.cpu().numpy().data).hexdigest()


# ---- arithmetic building blocks ---------------------------------------------------------------

@pytest.mark.parametrize("variant", [0, 1, 2])
def test_magnitude_mapping_exhaustive(pkg, torch_mod, variant):
    # every reachable gradient pair (gx, gy) in [-1020, 1020]^2: device mapping == min(255, isqrt(gx^2 + gy^2))
    assert pkg.hostapi.selftest_sobel_mag(variant) == 0


@pytest.mark.parametrize("mode_name,mode", [("IEEE", 0), ("FMA", 1)])
def test_all_2pow24_triples(pkg, oracle, known, torch_mod, mode_name, mode):
    """BASELINE config #2's shape: the 4096x4096 RGB cube holds every (r,g,b) exactly once."""
    cube = oracle.frame_cube()
    want = known[mode_name]["cube4096"]
    d = dev(torch_mod, cube)
    gray = pkg.hostapi.gray_to_plane(d, mode)
    assert sha(gray) == want["gray"]
    assert np.array_equal(gray.cpu().numpy(), oracle.gray_to_plane(cube, mode))
    assert sha(pkg.hostapi.gray_inplace(d.clone(), mode)) == want["grayrgb"]
    edge = pkg.hostapi.sobel(gray)
    assert sha(edge) == want["edge"]
    assert sha(pkg.hostapi.gray_sobel_fused(d, mode)) == want["edge"]


@pytest.mark.parametrize("name", ["smooth4096", "rand1_4096", "smooth1021x517"])
def test_known_answers_small(pkg, oracle, known, torch_mod, name):
    spec = known["frames"][name]
    frame = make_frame(oracle, name, spec)
    d = dev(torch_mod, frame)
    for mode_name, mode in (("IEEE", 0), ("FMA", 1)):
        want = known[mode_name][name]
        gray = pkg.hostapi.gray_to_plane(d, mode)
        assert sha(gray) == want["gray"], (name, mode_name)
        assert sha(pkg.hostapi.gray_inplace(d.clone(), mode)) == want["grayrgb"], (name, mode_name)
        edge = pkg.hostapi.sobel(gray)
        assert sha(edge) == want["edge"], (name, mode_name)
        if spec["w"] % 16 == 0:
            assert sha(pkg.hostapi.gray_sobel_fused(d, mode)) == want["edge"], (name, mode_name, "fused")
    if name in known["channel_sums"]:
        assert pkg.hostapi.sum_rgb_dev(d) == known["channel_sums"][name]


@pytest.mark.parametrize("name", ["rand1_16384", "smooth16384"])
def test_known_answers_full_size(pkg, oracle, known, torch_mod, name):
    """BASELINE configs #3/#4 frame: 16384x16384 RGB8, hashes from the compiled reference."""
    spec = known["frames"][name]
    frame = make_frame(oracle, name, spec)
    d = dev(torch_mod, frame)
    del frame
    want = known["IEEE"][name]
    gray = pkg.hostapi.gray_to_plane(d, 0)
    assert sha(gray) == want["gray"]
    edge = pkg.hostapi.sobel(gray)
    assert sha(edge) == want["edge"]
    assert int(edge.sum(dtype=torch_mod.int64).item()) == want["edge_sum"]
    fused = pkg.hostapi.gray_sobel_fused(d, 0)
    assert torch_mod.equal(fused, edge)
    del fused, edge
    # BASELINE configs[4]'s path at this size: the frame stays in HOST memory and goes through the band streamer (three hipStreams,
    # bands of 4096 and of 1000 rows -- the second does not divide the height) -- same known answer
    frame = make_frame(oracle, name, spec)
    for tile_rows in ((4096, 1000) if name == "rand1_16384" else (4096,)):
        edge_host = np.full((spec["h"], spec["w"]), 0x77, dtype=np.uint8)
        pkg.hostapi.gray_sobel_streamed(frame, edge_host, tile_rows=tile_rows)
        assert hashlib.sha256(edge_host.data).hexdigest() == want["edge"], tile_rows
    del frame, edge_host
    assert pkg.hostapi.sum_rgb_dev(d) == known["channel_sums"][name]
    # FMA mode at full size (hash of the plane only)
    assert sha(pkg.hostapi.gray_to_plane(d, 1, out=gray)) == known["FMA"][name]["gray"]
    # in place at full size, then size-independent properties: R==G==B==plane, idempotent from pass 2
    pkg.hostapi.gray_inplace(d, 0)
    assert sha(d) == want["grayrgb"]
    assert bool((d[..., 0] == d[..., 1]).all()) and bool((d[..., 0] == d[..., 2]).all())
    p2 = pkg.hostapi.gray_inplace(d.clone(), 0)
    p3 = pkg.hostapi.gray_inplace(p2.clone(), 0)
    assert torch_mod.equal(p2, p3)


def test_golden_image_crop(pkg, jpg_crop, torch_mod):
    """The reference's only golden fixture: images/test_gray.png (crop committed under tests/golden)."""
    d = dev(torch_mod, jpg_crop["rgb"])
    out = pkg.hostapi.gray_inplace(d.clone(), 0).cpu().numpy()
    assert np.array_equal(out, jpg_crop["golden_png"])
    assert np.array_equal(pkg.hostapi.gray_to_plane(d, 1).cpu().numpy(), jpg_crop["gray_fma"])
    edge = pkg.hostapi.sobel(dev(torch_mod, jpg_crop["gray_ieee"])).cpu().numpy()
    assert np.array_equal(edge[1:-1, 1:-1], jpg_crop["edge_ieee"][1:-1, 1:-1])
    assert not edge[0].any() and not edge[-1].any() and not edge[:, 0].any() and not edge[:, -1].any()


# ---- geometry edge cases ------------------------------------------------------------------------

SIZES = [(1, 1), (2, 2), (3, 3), (5, 2), (2, 9), (16, 3), (17, 4), (15, 15), (16, 16), (31, 33), (32, 34), (48, 1),
         (1024, 32), (1040, 33), (1008, 31), (2048, 65), (1023, 70), (1025, 3), (4096, 2), (333, 77)]


@pytest.mark.parametrize("w,h", SIZES)
def test_sobel_shapes(pkg, oracle, torch_mod, w, h):
    plane = oracle.frame_rand(w, h, 1, seed=w * 131 + h)[..., 0]
    got = pkg.hostapi.sobel(dev(torch_mod, plane)).cpu().numpy()
    assert np.array_equal(got, oracle.sobel(plane))


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("ch", [3, 4])
def test_gray_shapes(pkg, oracle, torch_mod, w, h, ch):
    img = oracle.frame_rand(w, h, ch, seed=w * 7 + h + ch)
    d = dev(torch_mod, img)
    for mode in (0, 1):
        assert np.array_equal(pkg.hostapi.gray_to_plane(d, mode).cpu().numpy(), oracle.gray_to_plane(img, mode))
        assert np.array_equal(pkg.hostapi.gray_inplace(d.clone(), mode).cpu().numpy(), oracle.gray_inplace(img, mode))
    assert np.array_equal(pkg.hostapi.extract_plane(d).cpu().numpy(), oracle.extract_plane(img))
    plane = oracle.frame_rand(w, h, 1, seed=99)[..., 0]
    assert np.array_equal(pkg.hostapi.replicate_plane(dev(torch_mod, plane), d.clone()).cpu().numpy(),
                          oracle.replicate_plane(plane, img))


def test_gray_channels_1_and_2(pkg, oracle, torch_mod):
    for ch in (1, 2):
        img = oracle.frame_rand(37, 11, ch, seed=ch)
        d = dev(torch_mod, img)
        assert np.array_equal(pkg.hostapi.gray_to_plane(d, 0).cpu().numpy(), img[..., 0])
        assert np.array_equal(pkg.hostapi.gray_inplace(d.clone(), 0).cpu().numpy(), img)   # documented no-op


def test_unaligned_device_pointers(pkg, oracle, torch_mod):
    """Buffers that are not 16-byte aligned take the per-pixel kernels; results are identical."""
    w, h = 64, 9
    img = oracle.frame_rand(w, h, 3, seed=5)
    raw = torch_mod.zeros(w * h * 3 + 64, dtype=torch_mod.uint8, device="cuda")
    for off in (1, 3, 8):
        view = raw[off:off + w * h * 3].view(h, w, 3)
        view.copy_(dev(torch_mod, img))
        assert view.data_ptr() % 16 != 0
        assert np.array_equal(pkg.hostapi.gray_to_plane(view, 0).cpu().numpy(), oracle.gray_to_plane(img, 0))
        assert np.array_equal(pkg.hostapi.gray_inplace(view, 0).cpu().numpy(), oracle.gray_inplace(img, 0))
    plane = oracle.frame_rand(w, h, 1, seed=6)[..., 0]
    rawp = torch_mod.zeros(w * h + 64, dtype=torch_mod.uint8, device="cuda")
    v = rawp[4:4 + w * h].view(h, w)
    v.copy_(dev(torch_mod, plane))
    assert np.array_equal(pkg.hostapi.sobel(v).cpu().numpy(), oracle.sobel(plane))


@pytest.mark.parametrize("w,h,parts", [(1024, 96, 3), (2048, 67, 2), (48, 10, 4), (1024, 64, 8), (100, 30, 3)])
def test_row_bands_equal_whole_frame(pkg, oracle, torch_mod, w, h, parts):
    """Row sharding on one device: N bands with halo rows == the unsharded result, byte for byte."""
    from importlib import import_module
    sharding = import_module("openmp-parallel-computing_amd.sharding")
    plane = oracle.frame_rand(w, h, 1, seed=w + h)[..., 0]
    want = oracle.sobel(plane)
    full = dev(torch_mod, plane)
    got = torch_mod.empty_like(full)
    for r, (y0, y1) in enumerate(sharding.row_partition(h, parts)):
        rows = y1 - y0
        if rows == 0:
            continue
        buf = torch_mod.zeros((rows + 2, w), dtype=torch_mod.uint8, device="cuda")
        buf[1:rows + 1] = full[y0:y1]
        if y0 > 0:
            buf[0] = full[y0 - 1]
        if y1 < h:
            buf[rows + 1] = full[y1]
        pkg.hostapi.sobel_rows(buf, got[y0:y1], rows, y0 == 0, y1 == h)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("w,h,ch", [(1024, 40, 3), (2064, 35, 3), (64, 5, 4), (16, 3, 3), (4096, 33, 4), (1040, 3, 3), (3072, 1, 3),
                                    (1024, 2, 3), (4112, 70, 3), (5136, 131, 3)])
def test_fused_equals_two_kernels(pkg, oracle, torch_mod, w, h, ch):
    img = oracle.frame_rand(w, h, ch, seed=w ^ h)
    d = dev(torch_mod, img)
    for mode in (0, 1):
        want = oracle.sobel(oracle.gray_to_plane(img, mode))
        assert np.array_equal(pkg.hostapi.gray_sobel_fused(d, mode).cpu().numpy(), want)
    # band form with RGB halo rows
    rows0 = h // 2
    if rows0 >= 1 and h - rows0 >= 1:
        want = oracle.sobel(oracle.gray_to_plane(img, 0))
        got = torch_mod.empty((h, w), dtype=torch_mod.uint8, device="cuda")
        for y0, y1 in ((0, rows0), (rows0, h)):
            rows = y1 - y0
            buf = torch_mod.zeros((rows + 2, w, ch), dtype=torch_mod.uint8, device="cuda")
            buf[1:rows + 1] = d[y0:y1]
            if y0 > 0:
                buf[0] = d[y0 - 1]
            if y1 < h:
                buf[rows + 1] = d[y1]
            pkg.hostapi.gray_sobel_fused_rows(buf, got[y0:y1], rows, 0, y0 == 0, y1 == h)
        assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("knobs", [{"GH_FUSED_DMA": "0"}, {"GH_FUSED_DMA": "2"}, {"GH_FUSED_DMA": "4"}, {"GH_FUSED_DMA": "3", "GH_FUSED_DMA_R": "16"},
                                   {"GH_FUSED_DMA": "3", "GH_FUSED_DMA_R": "64", "GH_FUSED_DMA_N0": "1", "GH_FUSED_DMA_R1": "32", "GH_FUSED_DMA_N1": "1",
                                    "GH_FUSED_DMA_R2": "16"}, {"GH_FUSED_DMA": "0", "GH_FUSED_VARIANT": "1"},
                                   {"GH_FUSED_DMA_R": "24", "GH_FUSED_DMA_ORDER": "2"}, {"GH_FUSED_DMA_R": "16", "GH_FUSED_DMA_ORDER": "1234"},
                                   {"GH_FUSED_DMA_R": "40", "GH_FUSED_DMA_ORDER": "1010"}, {"GH_FUSED_DMA_R": "8", "GH_FUSED_DMA_ORDER": "304"},
                                   {"GH_FUSED_DMA_R": "16", "GH_FUSED_DMA_ORDER": "0"}, {"GH_FUSED_DMA_R": "8", "GH_FUSED_DMA_ORDER": "36"}, {"GH_FUSED_DMA_R": "8", "GH_FUSED_DMA_ORDER": "8"}, {"GH_FUSED_DMA_R": "16", "GH_FUSED_DMA_ORDER": "2008"}, {"GH_FUSED_DMA_ORDER": "2006"},
                                   {"GH_FUSED_REG": "2", "GH_FUSED_DMA": "3", "GH_FUSED_DMA_R": "34"}, {"GH_FUSED_REG": "2", "GH_FUSED_DMA": "3", "GH_FUSED_DMA_R": "64"},
                                   {"GH_FUSED_REG": "2", "GH_FUSED_DMA": "3", "GH_FUSED_DMA_R": "22", "GH_FUSED_DMA_PAD_KB": "30"}])
@pytest.mark.lab
def test_fused_alternative_paths(pkg, oracle, torch_mod, knobs, monkeypatch):
    """Every tuning variant of the fused kernel (register-staged, LDS-DMA ring depths, band plans) is bit-exact too.
    tests/conftest.py sets GH_TUNE_DYNAMIC so that the library re-reads the knobs on every launch."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for (w, h, ch) in ((2064, 150, 3), (1024, 67, 3), (4096, 3, 3), (2064, 131, 4), (1024, 2, 4)):
        img = oracle.frame_rand(w, h, ch, seed=w + h)
        d = dev(torch_mod, img)
        for mode in (0, 1):
            want = oracle.sobel(oracle.gray_to_plane(img, mode))
            assert np.array_equal(pkg.hostapi.gray_sobel_fused(d, mode).cpu().numpy(), want), (knobs, w, h, ch, mode)


@pytest.mark.parametrize("knobs", [{"GH_SOBEL_DMA": "0"}, {"GH_SOBEL_DMA": "2"}, {"GH_SOBEL_DMA": "3"}, {"GH_SOBEL_DMA": "4"}, {"GH_SOBEL_DMA": "6"},
                                   {"GH_SOBEL_DMA": "8"}, {"GH_SOBEL_DMA": "6", "GH_SOBEL_R": "16"}, {"GH_SOBEL_DMA": "0", "GH_SOBEL_R": "32"},
                                   {"GH_SOBEL_R": "24", "GH_SOBEL_ORDER": "2"}, {"GH_SOBEL_R": "16", "GH_SOBEL_ORDER": "1234"},
                                   {"GH_SOBEL_R": "40", "GH_SOBEL_ORDER": "1010"}, {"GH_SOBEL_R": "128", "GH_SOBEL_ORDER": "4"},
                                   {"GH_SOBEL_R": "16", "GH_SOBEL_ORDER": "6"}, {"GH_SOBEL_R": "8", "GH_SOBEL_ORDER": "36"}, {"GH_SOBEL_R": "8", "GH_SOBEL_ORDER": "8"}, {"GH_SOBEL_R": "16", "GH_SOBEL_ORDER": "2008"}, {"GH_SOBEL_ORDER": "2000"}])
@pytest.mark.lab
def test_sobel_alternative_paths(pkg, oracle, torch_mod, knobs, monkeypatch):
    """Register-staged and LDS-DMA forms of the plane Sobel kernel, every ring depth: bit-exact, including bands
    (halo rows above / below) and the shifted last strip of widths that are not a multiple of 1024."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for (w, h) in ((2064, 150), (1024, 67), (4096, 3), (1040, 1), (3088, 2)):
        plane = oracle.frame_rand(w, h, 1, seed=w + 3 * h).reshape(h, w)
        want = oracle.sobel(plane)
        assert np.array_equal(pkg.hostapi.sobel(dev(torch_mod, plane)).cpu().numpy(), want), (knobs, w, h)
    # a middle band: rows above and below are real halo rows
    w, h = 2064, 40
    plane = oracle.frame_rand(w, h + 2, 1, seed=5).reshape(h + 2, w)
    want = oracle.sobel(plane)[1:-1]
    got = torch_mod.empty((h, w), dtype=torch_mod.uint8, device="cuda")
    pkg.hostapi.sobel_rows(dev(torch_mod, plane), got, h, False, False)
    assert np.array_equal(got.cpu().numpy(), want), knobs


# ---- the reference's drivers and historic kernels -----------------------------------------------

def test_pipeline_matches_reference_cli_fixtures(pkg, jpg_crop, cli_crop, torch_mod):
    rgb = jpg_crop["rgb"]
    rgba = np.dstack([rgb, cli_crop["rgba_alpha"]])
    for passes in (1, 2, 3):
        img, _, _ = pkg.hostapi.pipeline_gray_sobel_dev(dev(torch_mod, rgb), passes, 0)
        out = img.cpu().numpy()
        assert np.array_equal(out[..., 0], cli_crop[f"sobel_rgb_p{passes}_c0"]), passes
        assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
    for passes in (1, 2):
        img, _, _ = pkg.hostapi.pipeline_gray_sobel_dev(dev(torch_mod, rgba), passes, 0)
        out = img.cpu().numpy()
        assert np.array_equal(out[..., 0], cli_crop[f"sobel_rgba_p{passes}_c0"]), passes
        assert np.array_equal(out[..., 3], rgba[..., 3])


@pytest.mark.parametrize("w,h,ch,passes,mode", [(333, 77, 3, 1, 0), (333, 77, 3, 3, 1), (64, 48, 4, 2, 0), (1040, 40, 3, 2, 0)])
def test_pipeline_vs_oracle(pkg, oracle, torch_mod, w, h, ch, passes, mode):
    img = oracle.frame_rand(w, h, ch, seed=passes * 17 + w)
    want_img, _, want_edge = oracle.pipeline(img, passes, mode)
    got_img, _, got_edge = pkg.hostapi.pipeline_gray_sobel_dev(dev(torch_mod, img), passes, mode)
    assert np.array_equal(got_img.cpu().numpy(), want_img)
    assert np.array_equal(got_edge.cpu().numpy(), want_edge)


def test_host_pointer_shims(pkg, oracle, jpg_crop, cli_crop):
    """The reference-named entry points on HOST buffers (what main.c-style callers link against)."""
    rgb = jpg_crop["rgb"].copy()
    h, w, c = rgb.shape
    pkg.convert_to_grayscale(rgb, w, h, c)
    assert np.array_equal(rgb, jpg_crop["golden_png"])
    # the raw C symbol, void return, default mode
    lib = pkg.abi.load()
    raw = jpg_crop["rgb"].copy()
    lib.convert_to_grayscale(raw.ctypes.data, w, h, c)
    assert np.array_equal(raw, jpg_crop["golden_png"])
    # sobel_edge writes the interior only
    src = np.ascontiguousarray(jpg_crop["gray_ieee"])
    dst = np.full_like(src, 0x5A)
    pkg.sobel_edge(src, dst, w, h)
    assert (dst[0] == 0x5A).all() and (dst[-1] == 0x5A).all() and (dst[:, 0] == 0x5A).all() and (dst[:, -1] == 0x5A).all()
    assert np.array_equal(dst[1:-1, 1:-1], oracle.sobel(src)[1:-1, 1:-1])
    tiny = np.full((2, 7), 9, dtype=np.uint8)
    out = np.full((2, 7), 3, dtype=np.uint8)
    pkg.sobel_edge(tiny, out, 7, 2)                       # no interior: nothing written
    assert (out == 3).all()
    # multi-pass drivers
    img = jpg_crop["rgb"].copy()
    secs = pkg.hostapi.gray_passes(img, w, h, c, 2, 0)
    assert secs >= 0 and np.array_equal(img[..., 0], cli_crop["gray_rgb_p2_c0"])
    img = jpg_crop["rgb"].copy()
    pkg.hostapi.pipeline_gray_sobel(img, w, h, c, 2, 0)
    assert np.array_equal(img[..., 0], cli_crop["sobel_rgb_p2_c0"])
    rgba = np.ascontiguousarray(np.dstack([jpg_crop["rgb"], cli_crop["rgba_alpha"]]))
    pkg.convert_to_grayscale(rgba, w, h, 4)
    assert np.array_equal(rgba[..., 0], cli_crop["gray_rgba_p1_c0"]) and np.array_equal(rgba[..., 3], cli_crop["rgba_alpha"])


@pytest.mark.lab
@pytest.mark.parametrize("w,h,c,passes", [(2048, 1100, 3, 1), (2000, 1237, 4, 3), (4096, 1025, 3, 2)])
def test_host_shim_chunked_pipeline(pkg, oracle, w, h, c, passes, monkeypatch):
    """Large frames go through convert_to_grayscale's host entry point in chunks (an uploader thread copies chunk k+1 up while chunk
    k is converted and copied back); forced here on small frames: 1 Mpixel chunks, no size threshold."""
    monkeypatch.setenv("GH_HOST_CHUNK_MPX", "1")
    monkeypatch.setenv("GH_HOST_PIPELINE_MIN_MB", "0")
    img = oracle.frame_rand(w, h, c, seed=w + h + c)
    for mode in (0, 1):
        buf = img.copy()
        secs = pkg.hostapi.gray_passes(buf, w, h, c, passes, mode)
        assert secs > 0
        assert np.array_equal(buf, oracle.gray_passes(img, passes, mode)), (mode,)
    buf = img.copy()
    pkg.convert_to_grayscale(buf, w, h, c)                       # the reference-named symbol, default mode
    assert np.array_equal(buf, oracle.gray_inplace(img, 0))


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(2048, 1100), (1999, 1237), (4096, 1025), (16384, 200), (1024, 2049)])
def test_host_shim_chunked_sobel_edge(pkg, oracle, w, h, monkeypatch):
    """Large planes go through sobel_edge's host entry point in row chunks (upload of chunk k+1 overlaps the filter on the rows
    chunk k completed and their download); forced here on small planes: 1 Mpixel chunks (at least 64 rows), no size threshold.
    dst's border must stay untouched exactly as with the reference (sobel.c:11-12)."""
    monkeypatch.setenv("GH_HOST_CHUNK_MPX", "1")
    monkeypatch.setenv("GH_HOST_PIPELINE_MIN_MB", "0")
    plane = oracle.frame_rand(w, h, 1, seed=w + h).reshape(h, w)
    want = oracle.sobel(plane)
    dst = np.full((h, w), 0xA5, dtype=np.uint8)
    pkg.sobel_edge(plane, dst, w, h)
    assert np.array_equal(dst[1:-1, 1:-1], want[1:-1, 1:-1])
    assert (dst[0] == 0xA5).all() and (dst[-1] == 0xA5).all() and (dst[:, 0] == 0xA5).all() and (dst[:, -1] == 0xA5).all()
    monkeypatch.setenv("GH_HOST_PIPELINE_MIN_MB", "1000000")      # and the single-copy path gives the same bytes
    dst2 = np.full((h, w), 0xA5, dtype=np.uint8)
    pkg.sobel_edge(plane, dst2, w, h)
    assert np.array_equal(dst, dst2)


@pytest.mark.parametrize("name", ["rand1_1024x1024", "rand1_1021x517"])
def test_old_kernels(pkg, oracle, known, torch_mod, name):
    ka = known["old_kernels"][name]
    w, h = ka["w"], ka["h"]
    img = oracle.frame_rand(w, h, 3, seed=1)
    d = dev(torch_mod, img)
    assert pkg.hostapi.sum_rgb_dev(d) == ka["sums"]
    assert pkg.hostapi.sum_rgb(img) == ka["sums"]
    assert pkg.hostapi.avg_pixel(img) == ka["avg"]
    out, lo, hi = pkg.hostapi.mean_gray_minmax(d)
    assert [lo, hi] == ka["mean_gray_minmax"] and sha(out) == ka["mean_gray_rgb"]
    assert sha(pkg.hostapi.conv3x3(d, [1, 2, 1, 2, 4, 2, 1, 2, 1], 16)) == ka["conv_gauss16"]
    assert sha(pkg.hostapi.conv3x3(d, [1] * 9, 9)) == ka["conv_box9"]


def test_conv3x3_signed_kernel_and_small_frames(pkg, oracle, torch_mod):
    for w, h in ((1, 1), (2, 3), (17, 9), (64, 5)):
        img = oracle.frame_rand(w, h, 3, seed=w * h)
        d = dev(torch_mod, img)
        for k, norm in (([0, -1, 0, -1, 5, -1, 0, -1, 0], 1), ([-1, -2, -1, 0, 0, 0, 1, 2, 1], 3), ([1] * 9, 9)):
            assert np.array_equal(pkg.hostapi.conv3x3(d, k, norm).cpu().numpy(), oracle.conv3x3(img, k, norm)), (w, h, k)


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(342, 7), (343, 70), (344, 5), (350, 130), (1024, 67), (1365, 9), (1366, 33), (2731, 66), (5461, 20), (5464, 3), (683, 1), (700, 2)])
def test_conv3x3_ring_kernel_any_width_and_alignment(pkg, oracle, torch_mod, w, h, monkeypatch):
    """Rows of 1024 bytes and more (3 w >= 1024) run the LDS-DMA ring kernel k_conv3x3_dma whatever the width and the base
    alignment (aligned frames: plain pieces; rows that are multiples of 4 bytes: dword-shifted landing; others: + byte
    re-alignment); checked against the oracle, against the register-staged strips where those apply, for several kernels
    (signed taps, norm > 1) and input offsets 0..15; the bytes around the output must stay untouched."""
    t = torch_mod
    img = oracle.frame_rand(w, h, 3, seed=w * 7 + h)
    n = h * w * 3
    for k, norm in (([1, 2, 1, 2, 4, 2, 1, 2, 1], 16), ([0, -1, 0, -1, 5, -1, 0, -1, 0], 1), ([-1, -2, -1, 0, 0, 0, 1, 2, 1], 3), ([3, 1, 4, 1, 5, 9, 2, 6, 5], 36)):
        want = oracle.conv3x3(img, k, norm)
        for so in ((0, 1, 2, 3, 4, 7, 8, 12, 13, 15) if k[0] == 1 else (0, 5)):
            raw_in = t.zeros(n + 48, dtype=t.uint8, device="cuda")
            raw_out = t.full((n + 48,), 0x5A, dtype=t.uint8, device="cuda")
            src = raw_in[so:so + n].view(h, w, 3)
            do = (so * 5) % 16
            dst = raw_out[do:do + n].view(h, w, 3)
            src.copy_(dev(t, img))
            pkg.hostapi.conv3x3(src, k, norm, out=dst)
            assert np.array_equal(dst.cpu().numpy(), want), (k, norm, so)
            assert (raw_out[:do] == 0x5A).all() and (raw_out[do + n:] == 0x5A).all(), ("guard", so)
        monkeypatch.setenv("GH_CONV_DMA", "0")            # register-staged strips (aligned frames) / per-pixel kernel
        assert np.array_equal(pkg.hostapi.conv3x3(dev(t, img), k, norm).cpu().numpy(), want), (k, norm, "no dma")
        monkeypatch.delenv("GH_CONV_DMA")


SOG_SHAPES = [(16384, 130), (16384, 64), (8192, 192), (4096, 70), (2048, 16), (1920, 1080), (1024, 40), (1040, 33), (3072, 1), (2064, 3),
              (1031, 70), (2037, 100), (4099, 35), (16370, 40), (512, 20), (100, 30), (16, 3), (7, 5), (333, 77), (2, 2), (1, 9)]


@pytest.mark.parametrize("w,h", SOG_SHAPES)
@pytest.mark.parametrize("mode", [0, 1])
def test_sobel_of_gray_is_pass_k_of_the_reference_loop(pkg, oracle, torch_mod, w, h, mode):
    """gh_sobel_of_gray_u8_dev: pass k >= 2 of monolithic/src/main_with_sobel.c:51-62 in one launch -- the previous pass's edge plane e
    goes through lum(e, e, e) ON LOAD and then through the 3x3 window.  Against the oracle's convert_to_grayscale of (e, e, e) + plane
    extract + sobel_edge, for every kernel form the launcher picks (two strips / one strip, regular and generic bands, dword-shifted and
    byte-shifted rows, register-staged strips + column tail, per-pixel), both arithmetic modes.  The plane holds every grey level many
    times, the 13 (IEEE) / 13 (FMA) levels that lum() moves in particular."""
    t = torch_mod
    rs = np.random.RandomState(w * 31 + h)
    e = rs.randint(0, 256, (h, w)).astype(np.uint8)
    e[h // 2, : min(w, 256)] = np.arange(min(w, 256), dtype=np.uint8)          # every level next to its neighbours
    luma = oracle.gray_to_plane(np.repeat(e[:, :, None], 3, axis=2), mode)
    assert (luma != e).any() or w * h < 64
    want = oracle.sobel(luma)
    got = pkg.hostapi.sobel_of_gray(dev(t, e), mode)
    assert np.array_equal(got.cpu().numpy(), want), pkg.hostapi.last_kernel()
    assert "luma-on-load" in pkg.hostapi.last_kernel()
    # == the two launches it replaces
    two = pkg.hostapi.sobel(pkg.hostapi.gray_of_plane(dev(t, e), mode))
    assert t.equal(got, two)
    # row bands with halo rows: a partition of the frame == the whole frame
    if h >= 3:
        out = t.full((h, w), 0x5A, dtype=t.uint8, device="cuda")
        full = dev(t, e)
        for y0, y1 in pkg.sharding.row_partition(h, 3):
            if y1 == y0:
                continue
            top, bot = y0 == 0, y1 == h
            buf = t.full((y1 - y0 + 2, w), 0xEE, dtype=t.uint8, device="cuda")      # halo rows poisoned where the band edge is the frame border
            buf[1:y1 - y0 + 1] = full[y0:y1]
            if not top:
                buf[0] = full[y0 - 1]
            if not bot:
                buf[y1 - y0 + 1] = full[y1]
            pkg.hostapi.sobel_of_gray_rows(buf, out[y0:y1], y1 - y0, mode, top, bot)
        assert np.array_equal(out.cpu().numpy(), want)


def test_sobel_of_gray_full_size(pkg, oracle, known, torch_mod):
    """BASELINE's size: pass 2 of the pipeline on the 16384^2 known-answer frame.  The edge plane of pass 1 is pinned by the compiled
    reference's SHA-256; pass 2 in one launch must equal gray_of_plane + Sobel (both pinned kernels) bit for bit, in both modes, and
    run the two-strip ring kernel."""
    t = torch_mod
    img = oracle.frame_rand(16384, 16384, 3, seed=1)
    edge1 = pkg.hostapi.gray_sobel_fused(dev(t, img), 0)
    assert sha(edge1) == known["IEEE"]["rand1_16384"]["edge"]
    del img
    for mode in (0, 1):
        one = pkg.hostapi.sobel_of_gray(edge1, mode)
        assert pkg.hostapi.last_kernel() == "k_sobel_dma2 regular R=64 luma-on-load"
        two = pkg.hostapi.sobel(pkg.hostapi.gray_of_plane(edge1, mode))
        assert t.equal(one, two)
        assert not bool(one[0].any()) and not bool(one[-1].any()) and not bool(one[:, 0].any()) and not bool(one[:, -1].any())
        # and a band of it against the oracle
        e_np = edge1[5000:5066].cpu().numpy()
        want = oracle.sobel(oracle.gray_to_plane(np.repeat(e_np[:, :, None], 3, axis=2), mode))[1:-1]
        assert np.array_equal(one[5001:5065].cpu().numpy(), want)


def test_gray_of_plane_moves_13_levels(pkg, known, torch_mod):
    v = torch_mod.arange(256, dtype=torch_mod.uint8, device="cuda").repeat(5)
    out = pkg.hostapi.gray_of_plane(v, 0)[:256].cpu().numpy()
    moved = [int(i) for i in np.nonzero(out != np.arange(256))[0]]
    assert moved == known["gray_of_gray_not_fixed_ieee"]


def test_error_paths(pkg, torch_mod):
    t = torch_mod.zeros((4, 4), dtype=torch_mod.uint8, device="cuda")
    with pytest.raises(pkg.GrayHipError):
        pkg.hostapi.sobel(t, out=t)                       # src == dst
    with pytest.raises(pkg.GrayHipError):
        pkg.hostapi.gray_to_plane(torch_mod.zeros((2, 2, 5), dtype=torch_mod.uint8, device="cuda"))
    with pytest.raises(pkg.GrayHipError):
        pkg.hostapi.gray_sobel_fused(torch_mod.zeros((4, 16, 2), dtype=torch_mod.uint8, device="cuda"))    # RGB8 / RGBA8 only
    empty = torch_mod.zeros((0, 16, 3), dtype=torch_mod.uint8, device="cuda")
    assert pkg.hostapi.gray_to_plane(empty).numel() == 0


@pytest.mark.parametrize("world,w,h", [(1, 1024, 40), (3, 2048, 50), (4, 64, 13), (8, 1024, 64)])
def test_sharded_runners_on_one_device(pkg, oracle, torch_mod, world, w, h):
    """The multi-GPU runners (sharding.Sharded*), all ranks instantiated on ONE device, the RCCL halo exchange replaced
    by plain copies between the rank objects: exercises the pre-bound launchers, the interior / edge-row split and the
    border flags on real kernels.  (The exchange itself is covered with gloo in tests/test_sharding_gloo.py.)"""
    from importlib import import_module
    sharding = import_module("openmp-parallel-computing_amd.sharding")
    img = oracle.frame_rand(w, h, 3, seed=world * 1000 + h)
    want = oracle.sobel(oracle.gray_to_plane(img, 0))
    full = dev(torch_mod, img)
    backend = sharding.HipBackend(0)
    parts = sharding.row_partition(h, world)
    split = [sharding.ShardedGraySobel(sharding.ShardGeometry(r, world, w, *parts[r]), backend, "cuda") for r in range(world)]
    fused = [sharding.ShardedFusedGraySobel(sharding.ShardGeometry(r, world, w, *parts[r]), backend, "cuda") for r in range(world)]
    for r, (y0, y1) in enumerate(parts):
        fused[r].shard.copy_(full[y0:y1])

    def run(runner, halo_src, halo_buf):
        g = runner.g
        if g.rows == 0:
            return
        if g.world == 1:
            runner._whole()
            return
        if runner.overlap:
            runner._inner()
        # "exchange": neighbour rows straight from the source of truth
        if not g.top_is_border:
            halo_buf[0].copy_(halo_src[g.y0 - 1])
        if not g.bottom_is_border:
            halo_buf[g.rows + 1].copy_(halo_src[g.y1])
        if runner.overlap:
            runner._top()
            runner._bot()
        else:
            runner._whole()

    luma_full = pkg.hostapi.gray_to_plane(full, 0)
    got_split = torch_mod.empty((h, w), dtype=torch_mod.uint8, device="cuda")
    got_fused = torch_mod.empty_like(got_split)
    for r, (y0, y1) in enumerate(parts):
        if y1 == y0:
            continue
        pkg.hostapi.gray_to_plane(full[y0:y1], 0, out=split[r].luma[1:y1 - y0 + 1])
        run(split[r], luma_full, split[r].luma)
        run(fused[r], full, fused[r].rgb)
        got_split[y0:y1] = split[r].edge
        got_fused[y0:y1] = fused[r].edge
    assert np.array_equal(got_split.cpu().numpy(), want)
    assert np.array_equal(got_fused.cpu().numpy(), want)


@pytest.mark.parametrize("w,h,ch,tile_rows", [(1024, 100, 3, 32), (2048, 64, 3, 64), (64, 7, 4, 1), (4096, 333, 3, 50), (16, 3, 3, 2)])
def test_streamed_bands_equal_whole_frame(pkg, oracle, w, h, ch, tile_rows):
    """BASELINE configs[4] at test size: host frame -> bands over 3 hipStreams -> host edge plane == unstreamed result."""
    img = oracle.frame_rand(w, h, ch, seed=w + h + ch)
    edge = np.full((h, w), 0x77, dtype=np.uint8)
    secs = pkg.hostapi.gray_sobel_streamed(img, edge, tile_rows=tile_rows)
    assert secs > 0
    assert np.array_equal(edge, oracle.sobel(oracle.gray_to_plane(img, 0)))
    edge2 = np.zeros_like(edge)
    pkg.hostapi.gray_sobel_streamed(img, edge2, tile_rows=tile_rows, mode=1)
    assert np.array_equal(edge2, oracle.sobel(oracle.gray_to_plane(img, 1)))


@pytest.mark.parametrize("w,h", [(16, 1), (16, 5), (32, 33), (1024, 31), (1040, 40), (2048, 70), (4096, 9), (352, 100)])
def test_conv3x3_streaming_path(pkg, oracle, torch_mod, w, h):
    """old/parallel_convolution.c on u8 frames, strip-streaming kernel (3*w % 16 == 0): blur, sharpen, gradient kernels,
    negative and large divisors, and the fallback beyond the fast path's exactness conditions."""
    img = oracle.frame_rand(w, h, 3, seed=w * 3 + h)
    d = dev(torch_mod, img)
    cases = [([1, 2, 1, 2, 4, 2, 1, 2, 1], 16), ([1] * 9, 9), ([0, -1, 0, -1, 5, -1, 0, -1, 0], 1), ([-1, -2, -1, 0, 0, 0, 1, 2, 1], 3),
             ([1, 2, 3, 4, 5, 6, 7, 8, 9], 7), ([1, 2, 1, 2, 4, 2, 1, 2, 1], -16), ([-3, 0, 3, -10, 0, 10, -3, 0, 3], -2),
             ([100, 200, 100, 200, 400, 200, 100, 200, 100], 4096), ([7, 7, 7, 7, 7, 7, 7, 7, 7], 5000), ([0] * 9, 3),
             ([255, 255, 255, 255, 255, 255, 255, 255, 255], 1)]
    for k, norm in cases:
        got = pkg.hostapi.conv3x3(d, k, norm).cpu().numpy()
        assert np.array_equal(got, oracle.conv3x3(img, k, norm)), (w, h, k, norm)


# ---- native RCCL halo exchange (csrc/comm.hip) on one GPU: a world-of-1 communicator, ncclSend/ncclRecv to self --------

@pytest.fixture(scope="module")
def native_comm(pkg):
    comm = pkg.sharding.NativeHaloComm(0, 1)
    yield comm
    comm.close()


def test_native_comm_self_exchange(pkg, native_comm, torch_mod):
    t = torch_mod
    a = t.randint(0, 256, (49152,), dtype=t.uint8, device="cuda")
    b = t.randint(0, 256, (49152,), dtype=t.uint8, device="cuda")
    ra, rb = t.zeros_like(a), t.zeros_like(b)
    s = t.cuda.current_stream().cuda_stream
    native_comm.halo_exchange(0, a, ra, 0, b, rb, a.numel(), s)      # both "neighbours" are this rank
    t.cuda.synchronize()
    assert t.equal(ra, a) and t.equal(rb, b)
    native_comm.halo_exchange(-1, None, None, 0, b, ra, b.numel(), s)   # frame border above, neighbour below
    t.cuda.synchronize()
    assert t.equal(ra, b)
    sums = t.tensor([1, 2, 3], dtype=t.int64, device="cuda")
    native_comm.allreduce_sum_u64(sums, s)
    t.cuda.synchronize()
    assert sums.cpu().tolist() == [1, 2, 3]


@pytest.mark.parametrize("w,h", [(1024, 40), (2064, 3), (4096, 2), (16384, 130), (16384, 2048), (1031, 70), (8192, 128), (512, 20), (16384, 64), (4096, 1)])
@pytest.mark.parametrize("form", [2, 1])
@pytest.mark.parametrize("lb", LOOPBACK_MODES)
def test_native_sharded_steps_loopback(pkg, oracle, native_comm, torch_mod, w, h, form, lb):
    _sharded_steps_loopback(pkg, oracle, native_comm, torch_mod, w, h, form, lb)


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(1024, 40), (2064, 3), (16384, 130), (1031, 70), (512, 20)])
@pytest.mark.parametrize("lb", LOOPBACK_MODES)
def test_native_sharded_steps_loopback_event_form(pkg, oracle, lab_comm, torch_mod, w, h, lb):
    """form 0 (exchange || interior rows, then two edge-row launches behind an event): the slowest form measured, probe builds only"""
    _sharded_steps_loopback(pkg, oracle, lab_comm, torch_mod, w, h, 0, lb)


def _sharded_steps_loopback(pkg, oracle, native_comm, torch_mod, w, h, form, lb):
    """The C++ sharded step in its three forms -- 2 halo flag (default: exchange on the side stream beside ONE launch whose first /
    last band wait for a flag inside the kernel), 1 serial (exchange on the caller's stream, then one launch), 0 overlapped with
    events (exchange || interior rows, then first / last row) -- as each of the three kinds of shard a multi-GPU job has: a middle
    shard (lb 1), the top shard (lb 2: frame border above -- monolithic/src/sobel.c:11-12 -- and ONE halo row, below), the bottom shard
    (lb 3), the neighbours being this rank itself: a halo row becomes a copy of the shard's own edge row, so the result equals the
    oracle on the frame padded by replicating that row on the neighbour side only.  The halo rows start out poisoned and the steps
    are repeated, so a band that did not wait for its halo row shows -- and so does a border band that READ the row beyond the frame."""
    t = torch_mod
    img = oracle.frame_rand(w, h, 3, seed=w * 7 + h)
    want = loopback_sobel(oracle, oracle.gray_to_plane(img, 0), lb)
    up, down = lb in (1, 3), lb in (1, 2)
    native_comm.set_loopback(lb)
    native_comm.set_step_form(form)
    try:
        backend = pkg.sharding.HipBackend(0)
        # fused: RGB halo rows
        buf = t.full((h + 2, w, 3), 0xA5, dtype=t.uint8, device="cuda")
        buf[1:h + 1] = dev(t, img)
        edge = t.full((h, w), 7, dtype=t.uint8, device="cuda")
        step = native_comm.bind_fused_step(buf, edge, h, 0)
        step()                          # the very first step: halo rows still poisoned when the launch is enqueued
        t.cuda.synchronize()
        native_comm.check()
        assert np.array_equal(edge.cpu().numpy(), want)
        for _ in range(3):              # repeated steps: the exchange of step k+1 must wait for step k's edge rows
            buf[0] = 0x5A               # (enqueued on the same stream, in front of the step)
            buf[h + 1] = 0x5A
            step()
        t.cuda.synchronize()
        native_comm.check()
        assert np.array_equal(buf[0].cpu().numpy(), img[0] if up else np.full_like(img[0], 0x5A))       # border side: never written
        assert np.array_equal(buf[h + 1].cpu().numpy(), img[-1] if down else np.full_like(img[0], 0x5A))
        assert np.array_equal(edge.cpu().numpy(), want)
        # split: luma halo rows
        luma = t.full((h + 2, w), 0xA5, dtype=t.uint8, device="cuda")
        backend.gray_to_plane(dev(t, img), luma[1:h + 1])
        edge2 = t.full((h, w), 9, dtype=t.uint8, device="cuda")
        step2 = native_comm.bind_sobel_step(luma, edge2, h)
        step2()
        t.cuda.synchronize()
        assert np.array_equal(edge2.cpu().numpy(), want)
        luma[0] = 0x5A
        luma[h + 1] = 0x5A
        step2(); step2()
        t.cuda.synchronize()
        native_comm.check()
        assert np.array_equal(edge2.cpu().numpy(), want)
    finally:
        native_comm.set_step_form(-1)
        native_comm.set_loopback(0)


def test_native_pipelined_steps_loopback(pkg, oracle, native_comm, torch_mod):
    """Pipelined sharded step: frame k is processed while frame k+1's halo rows are exchanged; two different frames
    alternate, so a halo row that reached the wrong buffer (or arrived late) changes the result."""
    t = torch_mod
    w, h = 2064, 37
    frames = [oracle.frame_rand(w, h, 3, seed=s) for s in (11, 12, 13)]
    wants = []
    for img in frames:
        padded = np.concatenate([img[:1], img, img[-1:]], axis=0)
        e = oracle.sobel(oracle.gray_to_plane(padded, 0))[1:-1]
        e[:, 0] = 0
        e[:, -1] = 0
        wants.append(e)
    native_comm.set_loopback(True)
    try:
        bufs = []
        for img in frames:
            b = t.zeros((h + 2, w, 3), dtype=t.uint8, device="cuda")
            b[1:h + 1] = dev(t, img)
            bufs.append(b)
        edge = t.zeros((h, w), dtype=t.uint8, device="cuda")
        step = native_comm.bind_fused_step_pipelined(bufs, edge, h, 0)
        for k in range(6):
            step(k)
            t.cuda.synchronize()
            assert np.array_equal(edge.cpu().numpy(), wants[k % 3]), k
        # no synchronisation between steps: results of the last step only
        for k in range(6, 17):
            step(k)
        t.cuda.synchronize()
        assert np.array_equal(edge.cpu().numpy(), wants[16 % 3])
        # halo requested one frame ahead on a ring of two
        step_b = native_comm.bind_fused_step_pipelined(bufs[:2], edge, h, 0)
        for k in range(5):
            step_b(k)
        t.cuda.synchronize()
        assert np.array_equal(edge.cpu().numpy(), wants[4 % 2])
        # split pipeline: luma halo rows
        backend = pkg.sharding.HipBackend(0)
        lumas = []
        for img in frames:
            l = t.zeros((h + 2, w), dtype=t.uint8, device="cuda")
            backend.gray_to_plane(dev(t, img), l[1:h + 1])
            lumas.append(l)
        step2 = native_comm.bind_sobel_step_pipelined(lumas, edge, h)
        for k in range(5):
            step2(k)
        t.cuda.synchronize()
        assert np.array_equal(edge.cpu().numpy(), wants[4 % 3])
        # a single buffer (no prefetch possible) still works
        step1 = native_comm.bind_fused_step_pipelined(bufs[:1], edge, h, 0)
        step1(0); step1(1)
        t.cuda.synchronize()
        assert np.array_equal(edge.cpu().numpy(), wants[0])
    finally:
        native_comm.set_loopback(False)


@pytest.mark.parametrize("batch,nbuf", [(1, 2), (2, 4), (3, 6), (4, 8), (2, 8)])
def test_native_batched_halo_prefetch_loopback(pkg, oracle, native_comm, torch_mod, batch, nbuf):
    """Batched form of the pipelined step: every `batch` steps the halo rows of the next `batch` frames are exchanged in one group
    (gh_sharded_prefetch_halos_dev) and each step is a single launch.  All frames differ and their halo rows start out as garbage,
    so a halo that reached the wrong buffer, arrived late or was never exchanged changes the result; the frames are then
    overwritten with new pixels between two passes over the ring (the next batch request must pick the new rows up)."""
    t = torch_mod
    w, h = 2064, 37

    def expected(img):
        padded = np.concatenate([img[:1], img, img[-1:]], axis=0)       # loopback: both neighbours are this shard
        e = oracle.sobel(oracle.gray_to_plane(padded, 0))[1:-1]
        e[:, 0] = 0
        e[:, -1] = 0
        return e

    native_comm.set_loopback(True)
    try:
        frames = [oracle.frame_rand(w, h, 3, seed=100 + s) for s in range(nbuf)]
        bufs = []
        for img in frames:
            b = t.full((h + 2, w, 3), 0x77, dtype=t.uint8, device="cuda")
            b[1:h + 1] = dev(t, img)
            bufs.append(b)
        edge = t.zeros((h, w), dtype=t.uint8, device="cuda")
        step = native_comm.bind_fused_step_batched(bufs, edge, h, 0, batch)
        k = 0
        for _ in range(2 * nbuf + 1):                                   # synchronised: every step checked
            step(k)
            t.cuda.synchronize()
            assert np.array_equal(edge.cpu().numpy(), expected(frames[k % nbuf])), k
            k += 1
        while k % nbuf:                                                 # finish the pass over the ring
            step(k)
            k += 1
        t.cuda.synchronize()
        # new pixels in every buffer (enqueued on the same stream, as the contract asks), halo rows poisoned again
        frames = [oracle.frame_rand(w, h, 3, seed=500 + s) for s in range(nbuf)]
        for b, img in zip(bufs, frames):
            b[1:h + 1] = dev(t, img)
            b[0] = 0x33
            b[h + 1] = 0x44
        results = []
        for _ in range(2 * nbuf):                                       # unsynchronised: results copied out on the stream
            step(k)
            results.append((k, edge.clone()))
            k += 1
        t.cuda.synchronize()
        for kk, e in results[batch:]:            # (the first `batch` frames of this pass had been requested before the overwrite:
            assert np.array_equal(e.cpu().numpy(), expected(frames[kk % nbuf])), kk       # a caller refreshes frames BEFORE their batch request)
    finally:
        native_comm.set_loopback(False)


@pytest.mark.parametrize("w,h,c", [(1031, 40, 3), (2063, 70, 4), (1047, 3, 3), (33, 9, 3), (4099, 2, 4), (7, 5, 3), (1029, 66, 3), (2049, 17, 3),
                                   (1025, 5, 3), (1025, 67, 4), (1026, 130, 3), (3073, 65, 3), (2037, 90, 3), (16370, 24, 3),
                                   (2052, 40, 3), (1044, 9, 3), (16380, 24, 3), (5000, 33, 3), (1039, 66, 4), (1028, 35, 4), (1030, 7, 3)])
@pytest.mark.lab
@pytest.mark.parametrize("dma_unaligned", ["2", "1", "0"])
def test_fused_widths_that_are_not_multiples_of_16(pkg, oracle, torch_mod, w, h, c, dma_unaligned, monkeypatch):
    """The fused C entry point takes any width.  Widths >= 1024: the LDS-DMA ring kernel itself -- GH_FUSED_DMA_UNALIGNED=2
    (default): every row's pieces start at the 16-byte aligned address before the strip and land dword-shifted in LDS (rows
    that are multiples of 4 bytes: nothing else; other rows: v_alignbyte re-alignment); =1: pieces from the strip's own
    unaligned address; =0: register-staged unaligned strips + per-pixel column tail instead; narrower frames: strips + tail or
    the per-pixel kernel alone; unaligned buffer bases; whole frames and bands with real RGB halo rows."""
    monkeypatch.setenv("GH_FUSED_DMA_UNALIGNED", dma_unaligned)
    img = oracle.frame_rand(w, h, c, seed=w * 3 + h)
    d = dev(torch_mod, img)
    for min_px in ("1", str(1 << 30)):
        monkeypatch.setenv("GH_FUSED_UNALIGNED_MIN_PX", min_px)
        for mode in (0, 1):
            want = oracle.sobel(oracle.gray_to_plane(img, mode))
            assert np.array_equal(pkg.hostapi.gray_sobel_fused(d, mode).cpu().numpy(), want), (min_px, mode)
            if h >= 3:
                got = torch_mod.empty((h - 2, w), dtype=torch_mod.uint8, device="cuda")
                pkg.hostapi.gray_sobel_fused_rows(d, got, h - 2, mode, False, False)
                assert np.array_equal(got.cpu().numpy(), want[1:h - 1]), ("band", min_px, mode)
    # width a multiple of 16, buffers at odd addresses; guard bytes around the output must stay untouched
    monkeypatch.setenv("GH_FUSED_UNALIGNED_MIN_PX", "1")
    w2 = (w + 15) & ~15
    img2 = oracle.frame_rand(w2, h, c, seed=w2)
    raw_in = torch_mod.zeros(h * w2 * c + 32, dtype=torch_mod.uint8, device="cuda")
    raw_out = torch_mod.zeros(h * w2 + 32, dtype=torch_mod.uint8, device="cuda")
    src = raw_in[7:7 + h * w2 * c].view(h, w2, c)
    dst = raw_out[3:3 + h * w2].view(h, w2)
    src.copy_(dev(torch_mod, img2))
    pkg.hostapi.gray_sobel_fused(src, 0, out=dst)
    assert np.array_equal(dst.cpu().numpy(), oracle.sobel(oracle.gray_to_plane(img2, 0)))
    assert not raw_out[:3].any() and not raw_out[3 + h * w2:].any()


def test_gray_sobel_any_width(pkg, oracle, torch_mod):
    for (w, h, c) in ((1037, 40, 3), (1040, 40, 4), (19, 7, 3)):
        img = oracle.frame_rand(w, h, c, seed=w)
        assert np.array_equal(pkg.hostapi.gray_sobel(dev(torch_mod, img), 1).cpu().numpy(), oracle.sobel(oracle.gray_to_plane(img, 1)))


def test_streamed_frame_with_a_width_that_is_not_a_multiple_of_16(pkg, oracle):
    w, h = 1037, 150
    img = oracle.frame_rand(w, h, 3, seed=5)
    got = np.zeros((h, w), dtype=np.uint8)
    pkg.hostapi.gray_sobel_streamed(img, got, tile_rows=41)
    assert np.array_equal(got, oracle.sobel(oracle.gray_to_plane(img, 0)))


def test_streamed_row_blocks_equal_whole_frame(pkg, oracle):
    """A host frame split into row blocks (one per GPU in the 8-GPU streamed configuration): each block streamed on its own,
    halo rows read from host memory next to it, equals the whole-frame result."""
    w, h = 1040, 301
    img = oracle.frame_rand(w, h, 3, seed=77)
    want = oracle.sobel(oracle.gray_to_plane(img, 0))
    got = np.zeros((h, w), dtype=np.uint8)
    for (y0, y1) in pkg.sharding.row_partition(h, 3):
        part = np.zeros((y1 - y0, w), dtype=np.uint8)
        pkg.hostapi.gray_sobel_streamed_rows(img, part, y0, y1, tile_rows=37)
        got[y0:y1] = part
    assert np.array_equal(got, want)
    # a rank that holds only its own rows + halo rows
    y0, y1 = 100, 200
    local = np.ascontiguousarray(img[y0 - 1:y1 + 1])
    part = np.zeros((y1 - y0, w), dtype=np.uint8)
    pkg.hostapi.gray_sobel_streamed_rows(local, part, y0, y1, tile_rows=64, first_row_of_frame=y0 - 1, frame_height=h)
    assert np.array_equal(part, want[y0:y1])


@pytest.mark.parametrize("w,h", [(2048, 16), (2048, 64), (2064, 32), (4096, 128), (4112, 48), (6144, 64), (8192, 192), (8208, 16), (16384, 64), (16384, 80),
                                 (10240, 256), (16400, 128)])
@pytest.mark.lab
@pytest.mark.parametrize("band_rows", ["", "16", "64"])
def test_sobel_two_strips_per_wave(pkg, oracle, torch_mod, w, h, band_rows, monkeypatch):
    """k_sobel_dma2: a wave owns two neighbouring 1 KiB strips of every row (aligned frames whose rows split into whole bands of 64 or
    16 rows; the launcher picks it by itself for large frames).  Forced here (GH_SOBEL_NS=2) on small ones: widths that are and are
    not multiples of 2048 (the last pair of strips is shifted left), one and several bands, whole frames and bands with real halo
    rows; GH_SOBEL_NS=1 (one strip per wave) must give the same bytes."""
    if band_rows and h % int(band_rows):
        pytest.skip("rows do not split into whole bands of that height")
    monkeypatch.setenv("GH_SOBEL_NS", "2")
    if band_rows:
        monkeypatch.setenv("GH_SOBEL_R", band_rows)
    t = torch_mod
    plane = oracle.frame_rand(w, h + 2, 1, seed=w + h).reshape(h + 2, w)
    d = dev(t, plane)
    want = oracle.sobel(plane[1:h + 1])
    got = pkg.hostapi.sobel(d[1:h + 1])
    assert np.array_equal(got.cpu().numpy(), want)
    want_band = oracle.sobel(plane)[1:h + 1]                       # rows 1 .. h of the taller frame: real halo rows above and below
    for top, bot in ((False, False), (True, False), (False, True)):
        out = t.empty((h, w), dtype=t.uint8, device="cuda")
        pkg.hostapi.sobel_rows(d, out, h, top, bot)
        ref = want_band.copy()
        if top:
            ref[0] = 0
        if bot:
            ref[-1] = 0
        if top or bot:
            ref = oracle.sobel(plane[1 if top else 0:(h + 1) if bot else (h + 2)])
            ref = ref[(0 if top else 1):(0 if top else 1) + h]
        assert np.array_equal(out.cpu().numpy(), ref), (top, bot)
    monkeypatch.setenv("GH_SOBEL_NS", "1")
    assert np.array_equal(pkg.hostapi.sobel(d[1:h + 1]).cpu().numpy(), want)


def test_headline_shapes_run_the_kernels_they_are_meant_to_run(pkg, torch_mod):
    """gh_last_kernel(): the launch rules must send the headline shapes to the kernel forms DESIGN.md names -- a regression that silently
    fell back to a slower form (a changed threshold, a failed alignment test) would still be bit-exact and pass every parity test."""
    t, api = torch_mod, pkg.hostapi
    def sobel(w, h, off=0):
        raw = t.zeros(h * w + 64, dtype=t.uint8, device="cuda")
        api.sobel(raw[off:off + h * w].view(h, w))
        return api.last_kernel()
    def fused(w, h, c=3):
        api.gray_sobel_fused(t.zeros((h, w, c), dtype=t.uint8, device="cuda"), 0)
        return api.last_kernel()
    import ctypes
    cu, xcd = ctypes.c_int(0), ctypes.c_int(0)
    pkg.abi.check(pkg.abi.load().gh_device_topology(ctypes.byref(cu), ctypes.byref(xcd)), "gh_device_topology")
    assert (cu.value, xcd.value) == (256, 8), "the choices below are the rules' answers for an MI355X in SPX mode"
    assert sobel(16384, 16384) == "k_sobel_dma2 regular R=64"          # BASELINE configs[2]: two strips per wave, 64-row bands
    assert sobel(16384, 2048) == "k_sobel_dma2 regular R=16"           # the shard of 8 GPUs
    assert sobel(12288, 8192) == "k_sobel_dma regular R=16"            # off-step width: one strip, regular 16-row bands
    assert sobel(1920, 1080) == "k_sobel_dma regular R=16"             # the reference's test image: ragged last band shifted up
    assert sobel(16370, 16384) == "k_sobel_dma unaligned"              # unaligned rows, large: generic 64-row bands, aligned-down pieces
    assert sobel(16384, 16384, off=5) == "k_sobel_dma unaligned"
    assert sobel(2037, 1362) == "k_sobel_dma regular R=16 unaligned"
    assert sobel(640, 480) in ("k_sobel_stream", "k_sobel_tiled")      # narrower than one strip
    assert fused(16384, 16384) == "k_fused_dma D=2"                    # the bench's `value`
    assert fused(16384, 2048) == "k_fused_dma regular R=16"
    assert fused(16370, 16384) == "k_fused_dma D=2 unaligned"
    assert fused(1920, 1080) == "k_fused_dma regular R=16"
    assert fused(16384, 4096, 4) == "k_fused_dma D=3"
    api.conv3x3(t.zeros((4096, 16384, 3), dtype=t.uint8, device="cuda"), [1, 2, 1, 2, 4, 2, 1, 2, 1], 16)
    assert api.last_kernel() == "k_conv3x3_dma regular"
    api.conv3x3(t.zeros((100, 5000, 3), dtype=t.uint8, device="cuda"), [1] * 9, 9)
    assert api.last_kernel() == "k_conv3x3_dma regular unaligned"
    # pass k of the multi-pass loop: the plane Sobel's forms with the luma applied on load
    for (w, h), want in (((16384, 16384), "k_sobel_dma2 regular R=64"), ((16384, 2048), "k_sobel_dma2 regular R=16"), ((1920, 1080), "k_sobel_dma regular R=16"),
                         ((2037, 1362), "k_sobel_dma regular R=16 unaligned"), ((640, 480), "k_sobel_stream")):
        api.sobel_of_gray(t.zeros((h, w), dtype=t.uint8, device="cuda"), 0)
        assert api.last_kernel() == want + " luma-on-load"


def test_release_library_ignores_the_labs_knobs(pkg, oracle, known, torch_mod, monkeypatch):
    """The library callers get is the one this suite runs: no tuning knob is read.  With the lab's knobs set to values that would send a
    probe build to its slowest / probe-only forms, the release library launches the same kernels as without them and reproduces the
    compiled reference's known answers at BASELINE's 4096^2 size."""
    t, api = torch_mod, pkg.hostapi
    img = oracle.frame_rand(4096, 4096, 3, seed=1)
    d = dev(t, img)

    def run():
        names = []
        edge = api.gray_sobel_fused(d, 0)
        names.append(api.last_kernel())
        gray = api.gray_to_plane(d, 0)
        e2 = api.sobel(gray)
        names.append(api.last_kernel())
        api.conv3x3(d, [1, 2, 1, 2, 4, 2, 1, 2, 1], 16)
        names.append(api.last_kernel())
        api.sobel_of_gray(e2, 0)
        names.append(api.last_kernel())
        return names, sha(edge), sha(gray), sha(e2)

    plain = run()
    assert plain[0][0].startswith("k_fused_dma") and plain[0][1].startswith("k_sobel_dma") and plain[0][2].startswith("k_conv3x3_dma")
    assert plain[1] == plain[3] == known["IEEE"]["rand1_4096"]["edge"] and plain[2] == known["IEEE"]["rand1_4096"]["gray"]
    for k, v in {"GH_SOBEL_DMA": "0", "GH_FUSED_DMA": "0", "GH_SOBEL_NS": "4", "GH_SOBEL_R": "7", "GH_FUSED_DMA_R": "5", "GH_SOBEL_IMPL": "tile",
                 "GH_FUSED_IMPL": "tile", "GH_SOBEL_CORE": "i16", "GH_CONV_DMA": "0", "GH_CONV_IMPL": "px", "GH_SOBEL_ST": "0", "GH_FUSED_DMA_ST": "12",
                 "GH_NUM_CU_OVERRIDE": "8", "GH_BAND_ALIAS_RULE": "0", "GH_SOBEL_ORDER": "0", "GH_HALO_MEMOPS": "3"}.items():
        monkeypatch.setenv(k, v)
    assert run() == plain


@pytest.mark.lab
def test_release_and_probe_builds_pick_the_same_kernels(pkg, oracle, torch_mod):
    """One source tree, two builds: with no knob set the probe build (whose knobs are environment reads) and the release library (whose knobs
    are their defaults, folded at compile time) must choose the same kernel form for every shape -- and produce the same bytes."""
    t, api = torch_mod, pkg.hostapi
    rs = np.random.RandomState(20261005)
    shapes = [(16384, 2048), (16384, 130), (12288, 200), (8192, 4101), (4096, 4100), (2037, 1362), (1920, 1080), (16370, 200), (1031, 70), (640, 480), (100, 30), (7, 5)]
    shapes += [(int(rs.randint(1, 9000)), int(rs.randint(1, 300))) for _ in range(30)]
    lab_lib = pkg.abi.load()
    assert lab_lib._name == pkg.abi.LAB_LIB_PATH
    for w, h in shapes:
        img = t.from_numpy(oracle.frame_rand(w, h, 3, seed=w + h)).cuda()
        plane = img[:, :, 0].contiguous()
        got = {}
        for which in ("lab", "release"):
            ctx = pkg.abi.use_library(pkg.abi.LIB_PATH if which == "release" else pkg.abi.LAB_LIB_PATH)
            with ctx:
                api.init(0)
                res = []
                res.append((api.gray_sobel_fused(img, 0), api.last_kernel()))
                res.append((api.sobel(plane), api.last_kernel()))
                res.append((api.sobel_of_gray(plane, 0), api.last_kernel()))
                res.append((api.conv3x3(img, [1, 2, 1, 2, 4, 2, 1, 2, 1], 16), api.last_kernel()))
                got[which] = res
        for (a, ka), (b, kb) in zip(got["lab"], got["release"]):
            assert ka == kb, (w, h, ka, kb)
            assert t.equal(a, b), (w, h, ka)


@pytest.mark.lab
@pytest.mark.parametrize("w,h,band_rows", [(4096, 70, ""), (2048, 17, ""), (8192, 200, "64"), (8192, 65, "64"), (6160, 47, "16"), (2050, 35, "")])
def test_sobel_two_strips_ragged_last_band(pkg, oracle, torch_mod, w, h, band_rows, monkeypatch):
    """Rows that do not split into whole bands: k_sobel_dma2 shifts its last band up to end at the frame edge (the shared rows are
    computed twice).  (The two-launch split of very large frames is test_sobel_two_strips_split_into_two_launches.)"""
    monkeypatch.setenv("GH_SOBEL_NS", "2")
    if band_rows:
        monkeypatch.setenv("GH_SOBEL_R", band_rows)
    plane = oracle.frame_rand(w, h + 2, 1, seed=w + 5 * h).reshape(h + 2, w)
    d = dev(torch_mod, plane)
    assert np.array_equal(pkg.hostapi.sobel(d).cpu().numpy(), oracle.sobel(plane))
    out = torch_mod.empty((h, w), dtype=torch_mod.uint8, device="cuda")
    pkg.hostapi.sobel_rows(d, out, h, False, False)
    assert np.array_equal(out.cpu().numpy(), oracle.sobel(plane)[1:h + 1])


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(4096, 64), (4096, 130), (8192, 192), (16384, 64), (16384, 80), (4112, 64), (12288, 131), (5008, 70)])
@pytest.mark.parametrize("depth,core,band_rows", [("3", "1", ""), ("3", "0", ""), ("6", "1", ""), ("2", "1", "32"), ("3", "1", "16"), ("2", "0", "16")])
def test_sobel_four_strips_per_wave(pkg, oracle, torch_mod, w, h, depth, core, band_rows, monkeypatch):
    """k_sobel_dma4 (on request: GH_SOBEL_NS=4): a wave owns FOUR neighbouring 1 KiB strips of every row on the packed-i16 state, with
    the packed-f32 or the scalar magnitude; the pixels the strips exchange travel by v_readlane.  Whole frames (last band shifted up,
    last group of strips shifted left) and bands with real halo rows; frames too short for a band fall back to the other kernels."""
    monkeypatch.setenv("GH_SOBEL_NS", "4")
    monkeypatch.setenv("GH_SOBEL_NS4_D", depth)
    monkeypatch.setenv("GH_SOBEL_NS4_CORE", core)
    if band_rows:
        monkeypatch.setenv("GH_SOBEL_R", band_rows)
    plane = oracle.frame_rand(w, h + 2, 1, seed=w + 3 * h).reshape(h + 2, w)
    d = dev(torch_mod, plane)
    assert np.array_equal(pkg.hostapi.sobel(d).cpu().numpy(), oracle.sobel(plane))
    if h + 2 >= int(band_rows or 64):
        assert pkg.hostapi.last_kernel().startswith("k_sobel_dma4")
    out = torch_mod.full((h, w), 0x21, dtype=torch_mod.uint8, device="cuda")
    pkg.hostapi.sobel_rows(d, out, h, False, False)
    assert np.array_equal(out.cpu().numpy(), oracle.sobel(plane)[1:h + 1])
    pkg.hostapi.sobel_rows(d, out, h, True, False)
    assert np.array_equal(out.cpu().numpy(), oracle.sobel(plane[1:])[:h])


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(8192, 4101), (8192, 4111), (16384, 2050)])
def test_sobel_two_strips_split_into_two_launches(pkg, oracle, torch_mod, w, h, monkeypatch):
    """Frames whose whole bands fill the resident workgroup slots exactly and that have a few rows more: a shifted extra band would add
    a layer of workgroups, so k_sobel_dma2 takes the whole bands (bottom edge: a real halo row) and a second launch the ragged rest
    (top edge: a real halo row).  Production takes this path from 2^28 pixels up (16384 x 16390); GH_SOBEL_SPLIT_MIN_PX brings it
    down to a frame the oracle finishes in a second.  Checked as a whole frame and as a band with halo rows on both sides."""
    monkeypatch.setenv("GH_SOBEL_NS", "2")
    monkeypatch.setenv("GH_SOBEL_R", "16")
    monkeypatch.setenv("GH_SOBEL_SPLIT_MIN_PX", "1")
    plane = oracle.frame_rand(w, h + 2, 1, seed=w + 7 * h).reshape(h + 2, w)
    want = oracle.sobel(plane)
    d = dev(torch_mod, plane)
    assert np.array_equal(pkg.hostapi.sobel(d[1:h + 1].contiguous()).cpu().numpy(), oracle.sobel(plane[1:h + 1]))
    assert pkg.hostapi.last_kernel().startswith("k_sobel_dma")             # (the second launch: the one-strip kernel on the ragged rest)
    for top, bot in ((False, False), (True, False), (False, True)):
        out = torch_mod.full((h, w), 0x33, dtype=torch_mod.uint8, device="cuda")
        pkg.hostapi.sobel_rows(d, out, h, top, bot)
        ref = want[1:h + 1].copy()
        if top or bot:
            ref = oracle.sobel(plane[(1 if top else 0):(h + 1 if bot else h + 2)])[(0 if top else 1):][:h]
        assert np.array_equal(out.cpu().numpy(), ref), (top, bot)
    monkeypatch.setenv("GH_SOBEL_SPLIT_MIN_PX", str(1 << 30))              # the same frames without the split: shifted band or one-strip kernel
    assert np.array_equal(pkg.hostapi.sobel(d[1:h + 1].contiguous()).cpu().numpy(), oracle.sobel(plane[1:h + 1]))


@pytest.mark.lab
@pytest.mark.parametrize("w,h", [(2048, 16), (2049, 16), (2050, 32), (2051, 16), (2052, 48), (2060, 64), (4100, 32), (4099, 128), (6147, 16), (16370, 32), (16380, 64)])
def test_sobel_two_strips_per_wave_unaligned_rows(pkg, oracle, torch_mod, w, h, monkeypatch):
    """k_sobel_dma2 on frames whose rows are not 16-byte aligned (aligned-down pieces, dword-shifted landing, byte re-alignment of
    both strips and of the pixels they exchange), every input base offset 0..15, whole frames and a band with real halo rows;
    guard bytes around the output."""
    monkeypatch.setenv("GH_SOBEL_NS", "2")
    t = torch_mod
    plane = oracle.frame_rand(w, h + 2, 1, seed=w * 3 + h).reshape(h + 2, w)
    want = oracle.sobel(plane[1:h + 1])
    want_band = oracle.sobel(plane)[1:h + 1]
    n = (h + 2) * w
    for so in range(16):
        raw_in = t.zeros(n + 48, dtype=t.uint8, device="cuda")
        raw_out = t.full((h * w + 48,), 0x5A, dtype=t.uint8, device="cuda")
        src = raw_in[so:so + n].view(h + 2, w)
        do = (so * 7) % 16
        dst = raw_out[do:do + h * w].view(h, w)
        src.copy_(dev(t, plane))
        pkg.hostapi.sobel(src[1:h + 1], out=dst)
        assert np.array_equal(dst.cpu().numpy(), want), ("frame", so)
        assert (raw_out[:do] == 0x5A).all() and (raw_out[do + h * w:] == 0x5A).all(), ("guard", so)
        pkg.hostapi.sobel_rows(src, dst, h, False, False)
        assert np.array_equal(dst.cpu().numpy(), want_band), ("band", so)


@pytest.mark.parametrize("w", [1024, 1025, 1027, 1030, 1036, 1039, 1040, 2051, 2064, 3073])
def test_every_base_alignment_through_the_ring_kernels(pkg, oracle, torch_mod, w):
    """The aligned-down LDS-DMA rows (UNAL variants) take a different branch per (address mod 16) of a strip's first byte: piece
    start, LDS landing offset, which tail blocks are fetched, the byte re-alignment.  Every input base offset 0..15 (and two output
    offsets), for the plane Sobel and the fused kernel (RGB8 and RGBA8), whole frames and bands with real halo rows; the bytes
    around the output must stay untouched."""
    t = torch_mod
    h = 6
    plane = oracle.frame_rand(w, h, 1, seed=w).reshape(h, w)
    want_s = oracle.sobel(plane)
    imgs = {c: oracle.frame_rand(w, h, c, seed=w + c) for c in (3, 4)}
    want_f = {c: oracle.sobel(oracle.gray_to_plane(imgs[c], 0)) for c in (3, 4)}
    for so in range(16):
        for do in (0, 7):
            raw_in = t.zeros(h * w + 48, dtype=t.uint8, device="cuda")
            raw_out = t.full((h * w + 48,), 0x5A, dtype=t.uint8, device="cuda")
            src = raw_in[so:so + h * w].view(h, w)
            dst = raw_out[do:do + h * w].view(h, w)
            src.copy_(dev(t, plane))
            pkg.hostapi.sobel(src, out=dst)
            assert np.array_equal(dst.cpu().numpy(), want_s), ("sobel", so, do)
            assert (raw_out[:do] == 0x5A).all() and (raw_out[do + h * w:] == 0x5A).all(), ("sobel guard", so, do)
            band = t.full((h - 2, w), 0x5A, dtype=t.uint8, device="cuda")
            pkg.hostapi.sobel_rows(src, band, h - 2, False, False)
            assert np.array_equal(band.cpu().numpy(), want_s[1:h - 1]), ("sobel band", so, do)
            for c in (3, 4):
                raw_in = t.zeros(h * w * c + 48, dtype=t.uint8, device="cuda")
                raw_out.fill_(0x5A)
                src3 = raw_in[so:so + h * w * c].view(h, w, c)
                src3.copy_(dev(t, imgs[c]))
                pkg.hostapi.gray_sobel_fused(src3, 0, out=dst)
                assert np.array_equal(dst.cpu().numpy(), want_f[c]), ("fused", c, so, do)
                assert (raw_out[:do] == 0x5A).all() and (raw_out[do + h * w:] == 0x5A).all(), ("fused guard", c, so, do)
                pkg.hostapi.gray_sobel_fused_rows(src3, band, h - 2, 0, False, False)
                assert np.array_equal(band.cpu().numpy(), want_f[c][1:h - 1]), ("fused band", c, so, do)


def test_random_geometry_sweep(pkg, oracle, torch_mod):
    """Seeded sweep over frame geometries around the kernels' internal boundaries (1024-px strips, 64/32/16-row bands, the
    shifted last strip, one- and two-row frames) for the plane Sobel, the fused RGB8 / RGBA8 kernels and their band forms."""
    rnd = np.random.RandomState(20261004)
    shapes = [(1024, 1), (1024, 64), (1024, 65), (2048, 66), (1040, 63), (3072, 17), (16, 1), (32, 2), (1008, 40)]
    for k in range(int(os.environ.get("GH_SWEEP", "24"))):
        w = 16 * int(rnd.randint(1, 330))
        if k % 3 == 2:
            w = int(rnd.randint(1024, 5300))          # any width through the LDS-DMA ring kernels
        h = int(rnd.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 66, 67, 127, 129, 130, 200]))
        shapes.append((w, h))
    for (w, h) in shapes:
        for ch in (3, 4):
            img = oracle.frame_rand(w, h, ch, seed=w * 3 + h + ch)
            d = dev(torch_mod, img)
            mode = (w // 16 + h) & 1
            luma = oracle.gray_to_plane(img, mode)
            want = oracle.sobel(luma)
            assert np.array_equal(pkg.hostapi.gray_sobel_fused(d, mode).cpu().numpy(), want), ("fused", w, h, ch, mode)
            if ch == 3:
                assert np.array_equal(pkg.hostapi.sobel(dev(torch_mod, luma)).cpu().numpy(), want), ("sobel", w, h)
            if h >= 3:      # a middle band: first and last row of the frame act as halo rows
                got = torch_mod.empty((h - 2, w), dtype=torch_mod.uint8, device="cuda")
                pkg.hostapi.gray_sobel_fused_rows(d, got, h - 2, mode, False, False)
                # rows 1..h-2 of the whole-frame result are exactly what a band with real halo rows must produce
                assert np.array_equal(got.cpu().numpy(), want[1:h - 1]), ("band", w, h, ch)


def test_device_entry_points_are_graph_capturable(pkg, oracle, torch_mod):
    """INTEGRATION.md promises that the *_dev entry points are asynchronous, allocation-free and capturable: record
    gray -> Sobel and the fused kernel into one hipGraph, replay it on fresh output buffers."""
    t = torch_mod
    w, h = 2064, 70
    img = oracle.frame_rand(w, h, 3, seed=3)
    want = oracle.sobel(oracle.gray_to_plane(img, 0))
    d = dev(t, img)
    luma = t.zeros((h, w), dtype=t.uint8, device="cuda")
    e1 = t.zeros((h, w), dtype=t.uint8, device="cuda")
    e2 = t.zeros((h, w), dtype=t.uint8, device="cuda")
    s = t.cuda.Stream()
    s.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(s):
        pkg.hostapi.gray_to_plane(d, 0, out=luma); pkg.hostapi.sobel(luma, out=e1); pkg.hostapi.gray_sobel_fused(d, 0, out=e2)   # warm-up
    t.cuda.current_stream().wait_stream(s)
    t.cuda.synchronize()
    g = t.cuda.CUDAGraph()
    with t.cuda.graph(g, stream=s):
        pkg.hostapi.gray_to_plane(d, 0, out=luma)
        pkg.hostapi.sobel(luma, out=e1)
        pkg.hostapi.gray_sobel_fused(d, 0, out=e2)
    for _ in range(3):
        luma.zero_(); e1.zero_(); e2.zero_()
        g.replay()
        t.cuda.synchronize()
        assert np.array_equal(e1.cpu().numpy(), want) and np.array_equal(e2.cpu().numpy(), want)


@pytest.mark.parametrize("w,h,c", [(200, 121, 3), (333, 77, 4), (1024, 64, 3), (17, 5, 3)])
def test_many_passes(pkg, oracle, torch_mod, w, h, c):
    """The reference's bench script runs its gray driver with hundreds of passes (bench_and_plot_monolithic.sh): here all of
    them run in ONE launch (per-pixel independent, gh_gray_inplace_passes_dev).  Gray and gray+Sobel drivers must equal N
    sequential passes of the oracle, bit for bit."""
    img = oracle.frame_rand(w, h, c, seed=w + h + c)
    for mode in (0, 1):
        for passes in (1, 2, 5, 40, 1000):
            want = oracle.gray_passes(img, passes, mode)
            host = img.copy()
            pkg.hostapi.gray_passes(host, w, h, c, passes, mode)
            assert np.array_equal(host, want), ("gray host", passes, mode)
            d = dev(torch_mod, img)
            pkg.hostapi.gray_inplace(d, mode, passes=passes)
            assert np.array_equal(d.cpu().numpy(), want), ("gray dev", passes, mode)
    for passes in (7, 8, 9, 70, 131):
        want_img, _, _ = oracle.pipeline(img, passes, 0)
        host = img.copy()
        pkg.hostapi.pipeline_gray_sobel(host, w, h, c, passes, 0)
        assert np.array_equal(host, want_img), ("pipeline", passes)


@pytest.mark.parametrize("w,h", [(1031, 40), (2063, 70), (1040 + 7, 3), (33, 9), (4099, 2), (1025, 5), (1026, 67), (1027, 130), (1028, 3),
                                 (2049, 70), (3073, 65), (2037, 90), (16370, 24), (2052, 40), (1044, 9), (16380, 24), (5000, 33), (1039, 66)])
@pytest.mark.lab
@pytest.mark.parametrize("dma_unaligned", ["2", "1", "0"])
def test_sobel_widths_that_are_not_multiples_of_16(pkg, oracle, torch_mod, w, h, dma_unaligned, monkeypatch):
    """Frame widths that are not a multiple of 16 (rows start at unaligned bytes).  Widths >= 1024: the LDS-DMA ring kernel --
    GH_SOBEL_DMA_UNALIGNED=2 (default): every row's piece starts at the 16-byte aligned address at or before the strip and
    lands dword-shifted in LDS (widths that are multiples of 4: nothing else; other widths: v_alignbyte re-alignment);
    =1: pieces from the strip's own unaligned address (the right halo dword of a strip that ends within 3 bytes of the row end
    is fetched earlier and shifted: w % 1024 in 1..3); =0 or narrower frames: the register-staged strip kernel on the first
    w & ~15 columns with unaligned vector accesses + a per-pixel kernel for the rest (forced on for small frames here)."""
    monkeypatch.setenv("GH_SOBEL_DMA_UNALIGNED", dma_unaligned)
    monkeypatch.setenv("GH_SOBEL_UNALIGNED_MIN_PX", "1")
    plane = oracle.frame_rand(w, h, 1, seed=w + h).reshape(h, w)
    want = oracle.sobel(plane)
    assert np.array_equal(pkg.hostapi.sobel(dev(torch_mod, plane)).cpu().numpy(), want)
    if h >= 3:    # middle band with real halo rows
        got = torch_mod.empty((h - 2, w), dtype=torch_mod.uint8, device="cuda")
        pkg.hostapi.sobel_rows(dev(torch_mod, plane), got, h - 2, False, False)
        assert np.array_equal(got.cpu().numpy(), want[1:h - 1])
    # width a multiple of 16 but planes that start at odd bytes
    w2 = (w + 15) & ~15
    plane2 = oracle.frame_rand(w2, h, 1, seed=w2 + h).reshape(h, w2)
    raw_in = torch_mod.zeros(h * w2 + 32, dtype=torch_mod.uint8, device="cuda")
    raw_out = torch_mod.zeros(h * w2 + 32, dtype=torch_mod.uint8, device="cuda")
    src = raw_in[5:5 + h * w2].view(h, w2)
    dst = raw_out[9:9 + h * w2].view(h, w2)
    src.copy_(dev(torch_mod, plane2))
    pkg.hostapi.sobel(src, out=dst)
    assert np.array_equal(dst.cpu().numpy(), oracle.sobel(plane2))
    assert not raw_out[:9].any() and not raw_out[9 + h * w2:].any()          # nothing written outside the plane
    monkeypatch.setenv("GH_SOBEL_UNALIGNED", "0")          # and the all-per-pixel path still agrees
    assert np.array_equal(pkg.hostapi.sobel(dev(torch_mod, plane)).cpu().numpy(), want)
