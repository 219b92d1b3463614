"""The drop-in CLIs' image codec (csrc/cli/image_io.cpp, jpeg_decode.cpp), exercised without a GPU through
bin/gh_imgtool.  The reference decodes with the stb_image.h it vendors (monolithic/src/main.c:21); its JPEG
output differs from libjpeg/PIL by up to +-3, so the checks here are against digests of the reference's own
decoder output (tests/golden/test_jpg_full.json, written by make_fixtures.py) and, in the build container,
against that decoder compiled in place (oracle/_ref/stb_decode)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, ROOT

TOOL = os.path.join(ROOT, "bin", "gh_imgtool")
REF_IMAGES = "/root/reference/images"
STB = os.path.join(ROOT, "oracle", "_ref", "stb_decode")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.access(TOOL, os.X_OK):
        subprocess.run(["make", "-C", ROOT, "bin/gh_imgtool"], check=True, capture_output=True)


def decode(path, tmp_path):
    raw = tmp_path / "d.raw"
    p = subprocess.run([TOOL, "decode", str(path), str(raw)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    w, h, c = map(int, p.stdout.split())
    return np.fromfile(raw, dtype=np.uint8).reshape(h, w, c)


def encode(arr, path, tmp_path):
    h, w = arr.shape[:2]
    c = 1 if arr.ndim == 2 else arr.shape[2]
    raw = tmp_path / "e.raw"
    np.ascontiguousarray(arr).tofile(raw)
    p = subprocess.run([TOOL, "encode", str(w), str(h), str(c), str(raw), str(path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_jpeg_matches_the_reference_decoder_digest(tmp_path):
    full = json.load(open(os.path.join(GOLDEN, "test_jpg_full.json")))
    img = decode(os.path.join(GOLDEN, "test.jpg"), tmp_path)
    assert img.shape == (1080, 1920, 3)
    assert hashlib.sha256(img.data).hexdigest() == full["decoded_rgb_sha256"]


@pytest.mark.skipif(not (os.path.isdir(REF_IMAGES) and os.access(STB, os.X_OK)), reason="reference tree only in the build container")
@pytest.mark.parametrize("name", ["test.jpg", "half_of_a_mega_photo.jpg", "more_than_one_mega_photo.jpg", "test_gray.png"])
def test_decoder_equals_reference_stb(tmp_path, name):
    """baseline 4:4:4, progressive 4:2:0 with odd width, and an RGB PNG: pixel-identical to the reference's decoder."""
    mine = decode(os.path.join(REF_IMAGES, name), tmp_path)
    raw = tmp_path / "s.raw"
    out = subprocess.run([STB, os.path.join(REF_IMAGES, name), str(raw)], check=True, capture_output=True, text=True).stdout.split()
    w, h, c = map(int, out)
    theirs = np.fromfile(raw, dtype=np.uint8).reshape(h, w, c)
    assert mine.shape == theirs.shape and np.array_equal(mine, theirs)


def _photo_like(w, h, seed):
    """Smooth gradients + edges + noise: every coefficient band and both chroma planes carry something."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 7) % 256], axis=-1).astype(np.int32)
    img[h // 3: 2 * h // 3, w // 4: w // 2] = [250, 20, 30]
    img += rng.randint(-25, 26, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


JPEG_VARIANTS = [
    # (name, size, PIL mode, save options)
    ("base444", (67, 45), "RGB", dict(quality=90, subsampling=0)),
    ("base422", (67, 45), "RGB", dict(quality=85, subsampling=1)),
    ("base420", (67, 45), "RGB", dict(quality=75, subsampling=2)),
    ("base420_tiny", (1, 1), "RGB", dict(quality=75, subsampling=2)),
    ("base420_narrow", (3, 50), "RGB", dict(quality=75, subsampling=2)),
    ("base420_16", (16, 16), "RGB", dict(quality=60, subsampling=2)),
    ("base420_17", (17, 33), "RGB", dict(quality=60, subsampling=2)),
    ("optimised", (130, 71), "RGB", dict(quality=92, subsampling=2, optimize=True)),
    ("q100", (64, 64), "RGB", dict(quality=100, subsampling=0)),
    ("q5", (120, 90), "RGB", dict(quality=5, subsampling=2)),
    ("prog444", (67, 45), "RGB", dict(quality=90, subsampling=0, progressive=True)),
    ("prog422", (131, 77), "RGB", dict(quality=80, subsampling=1, progressive=True)),
    ("prog420", (131, 77), "RGB", dict(quality=70, subsampling=2, progressive=True)),
    ("prog420_q30", (200, 120), "RGB", dict(quality=30, subsampling=2, progressive=True, optimize=True)),
    ("grey", (75, 51), "L", dict(quality=85)),
    ("grey_prog", (75, 51), "L", dict(quality=85, progressive=True)),
    ("restart_blocks", (150, 70), "RGB", dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    ("restart_rows", (150, 70), "RGB", dict(quality=80, subsampling=0, restart_marker_rows=1)),
    ("restart_prog", (150, 70), "RGB", dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=5)),
    ("cmyk", (60, 40), "CMYK", dict(quality=85)),
    ("big_prog", (517, 389), "RGB", dict(quality=88, subsampling=2, progressive=True)),
]


@pytest.mark.skipif(not os.access(STB, os.X_OK), reason="the reference's decoder is compiled only where /root/reference exists")
@pytest.mark.parametrize("name,size,mode,opts", JPEG_VARIANTS, ids=[v[0] for v in JPEG_VARIANTS])
def test_generated_jpegs_decode_like_the_reference(tmp_path, name, size, mode, opts):
    """Streams written here by libjpeg (through PIL) in the shapes a JPEG can take -- baseline / progressive, 4:4:4 / 4:2:2 /
    4:2:0, greyscale, restart intervals, optimised Huffman tables, Adobe CMYK -- must come out of this repo's decoder exactly as
    out of the reference's (stb_image v2.30 compiled in place): same size, same channel count, same bytes."""
    w, h = size
    rgb = _photo_like(w, h, seed=len(name) * 131 + w)
    if mode == "L":
        img = Image.fromarray(rgb[..., 0])
    elif mode == "CMYK":
        img = Image.fromarray(rgb).convert("CMYK")
    else:
        img = Image.fromarray(rgb)
    path = tmp_path / f"{name}.jpg"
    try:
        img.save(path, "JPEG", **opts)
    except TypeError:
        pytest.skip("this Pillow does not know one of the save options")
    mine = decode(path, tmp_path)
    raw = tmp_path / "s.raw"
    out = subprocess.run([STB, str(path), str(raw)], check=True, capture_output=True, text=True).stdout.split()
    sw, sh, sc = map(int, out)
    theirs = np.fromfile(raw, dtype=np.uint8).reshape(sh, sw, sc)
    assert mine.shape == theirs.shape, (mine.shape, theirs.shape)
    assert np.array_equal(mine, theirs), f"{name}: {np.count_nonzero(mine != theirs)} bytes differ"


STB_ENC = os.path.join(ROOT, "oracle", "_ref", "stb_encode")


def test_png_file_equals_the_reference_golden_file(tmp_path, oracle):
    """File-level parity: the reference's committed images/test_gray.png is the IEEE grey of images/test.jpg
    written by stb_image_write; this repo's decoder + the (oracle's) grey + this repo's PNG writer give the same FILE -- the
    md5 the survey recorded.  No GPU involved: the GPU form of this check is tests/test_cli_contract.py."""
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["test_jpg"]
    rgb = decode(os.path.join(GOLDEN, "test.jpg"), tmp_path)
    grey = oracle.gray_inplace(rgb, oracle.MODE_IEEE)
    path = tmp_path / "g.png"
    encode(grey, path, tmp_path)
    assert hashlib.md5(path.read_bytes()).hexdigest() == ka["golden_png_md5"]


@pytest.mark.skipif(not os.access(STB_ENC, os.X_OK), reason="the reference's PNG writer is compiled only where /root/reference exists")
@pytest.mark.parametrize("w,h,c,kind", [(1, 1, 1, "noise"), (2, 3, 3, "noise"), (7, 5, 2, "smooth"), (53, 37, 3, "smooth"), (64, 64, 4, "noise"),
                                        (300, 2, 3, "flat"), (2, 300, 1, "smooth"), (640, 480, 3, "photo"), (1000, 3, 4, "photo"),
                                        (257, 129, 1, "noise"), (33000, 1, 1, "flat"), (40, 900, 3, "flat")])
def test_png_files_equal_the_reference_writer(tmp_path, w, h, c, kind):
    """Byte-identical PNG files with the reference's writer (stb_image_write v1.16 compiled in place) on frames that exercise
    every branch of its compressor: long runs (length 258, distances up to the window), incompressible noise (stored-block
    fallback), every filter type, 1..4 channels, the 3-byte tail."""
    rng = np.random.RandomState(w * 7 + h * 3 + c)
    shape = (h, w, c) if c > 1 else (h, w)
    if kind == "noise":
        arr = rng.randint(0, 256, size=shape, dtype=np.uint8)
    elif kind == "flat":
        arr = np.full(shape, 200, dtype=np.uint8)
        arr.reshape(-1)[:: max(1, arr.size // 11)] = 3
    elif kind == "smooth":
        arr = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] % 256).astype(np.uint8)
        arr = np.repeat(arr, c, axis=2) if c > 1 else arr[..., 0]
    else:
        arr = _photo_like(w, h, seed=w + h)[..., :c] if c <= 3 else np.concatenate([_photo_like(w, h, seed=w + h), np.full((h, w, 1), 255, np.uint8)], axis=2)
    arr = np.ascontiguousarray(arr)
    mine, theirs = tmp_path / "mine.png", tmp_path / "theirs.png"
    encode(arr, mine, tmp_path)
    raw = tmp_path / "r.raw"
    arr.tofile(raw)
    subprocess.run([STB_ENC, str(w), str(h), str(c), str(raw), str(theirs)], check=True, capture_output=True)
    assert mine.read_bytes() == theirs.read_bytes()


def test_fast_png_mode_writes_the_same_pixels(tmp_path, monkeypatch):
    arr = _photo_like(333, 211, seed=5)
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    encode(arr, a, tmp_path)
    monkeypatch.setenv("GRAYHIP_PNG", "fast")
    encode(arr, b, tmp_path)
    assert a.read_bytes() != b.read_bytes()
    assert np.array_equal(np.asarray(Image.open(a)), arr) and np.array_equal(np.asarray(Image.open(b)), arr)


@pytest.mark.parametrize("w,h,c", [(1, 1, 1), (7, 3, 2), (53, 37, 3), (64, 64, 4), (300, 2, 3), (2, 300, 1)])
def test_png_encode_roundtrip(tmp_path, w, h, c):
    rng = np.random.RandomState(w * 100 + h + c)
    arr = rng.randint(0, 256, size=(h, w, c) if c > 1 else (h, w), dtype=np.uint8)
    if c >= 3:   # some smooth structure so every filter type gets picked
        arr[: h // 2] = (np.add.outer(np.arange(h // 2), np.arange(w))[..., None] % 256).astype(np.uint8)
    path = tmp_path / "o.png"
    encode(arr, path, tmp_path)
    back = np.asarray(Image.open(path))
    assert back.shape == arr.shape and np.array_equal(back, arr)           # PIL reads what we wrote
    mine = decode(path, tmp_path)
    assert np.array_equal(mine.reshape(arr.shape), arr)                    # and so do we


def test_png_decode_variants(tmp_path):
    rng = np.random.RandomState(5)
    rgb = rng.randint(0, 256, size=(33, 47, 3), dtype=np.uint8)
    # interlaced RGB / RGBA / grey / grey+alpha / palette (+tRNS) / 16-bit / low bit depths
    Image.fromarray(rgb).save(tmp_path / "a.png")
    assert np.array_equal(decode(tmp_path / "a.png", tmp_path), rgb)
    rgba = np.dstack([rgb, rng.randint(0, 256, size=(33, 47), dtype=np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "b.png")
    assert np.array_equal(decode(tmp_path / "b.png", tmp_path), rgba)
    grey = rgb[..., 0]
    Image.fromarray(grey, "L").save(tmp_path / "c.png")
    assert np.array_equal(decode(tmp_path / "c.png", tmp_path)[..., 0], grey)
    la = np.dstack([grey, rgb[..., 1]])
    Image.fromarray(la, "LA").save(tmp_path / "d.png")
    assert np.array_equal(decode(tmp_path / "d.png", tmp_path), la)
    pal = Image.fromarray(rgb).quantize(colors=37)
    pal.save(tmp_path / "e.png")
    assert np.array_equal(decode(tmp_path / "e.png", tmp_path), np.asarray(pal.convert("RGB")))
    one = Image.fromarray((grey > 127).astype(np.uint8) * 255, "L").convert("1")
    one.save(tmp_path / "f.png")
    assert np.array_equal(decode(tmp_path / "f.png", tmp_path)[..., 0], np.asarray(one.convert("L")))
    g16 = (rng.randint(0, 65536, size=(9, 11)).astype(np.uint16))
    Image.fromarray(g16, "I;16").save(tmp_path / "g.png")
    assert np.array_equal(decode(tmp_path / "g.png", tmp_path)[..., 0], (g16 >> 8).astype(np.uint8))   # stb: high byte


def _adam7_png(arr: np.ndarray) -> bytes:
    """Minimal Adam7-interlaced 8-bit PNG writer (filter 0 everywhere) -- PIL cannot write interlaced files."""
    import struct
    import zlib
    h, w = arr.shape[:2]
    c = 1 if arr.ndim == 2 else arr.shape[2]
    a = arr.reshape(h, w, c)
    xs, ys, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    raw = b""
    for p in range(7):
        sub = a[ys[p]::dy[p], xs[p]::dx[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 1))
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("w,h,c", [(1, 1, 3), (3, 2, 3), (8, 8, 1), (13, 21, 4), (47, 33, 3)])
def test_png_adam7_interlaced(tmp_path, w, h, c):
    rng = np.random.RandomState(w + 10 * h)
    arr = rng.randint(0, 256, size=(h, w, c), dtype=np.uint8)
    (tmp_path / "i.png").write_bytes(_adam7_png(arr))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "i.png")).reshape(h, w, c), arr)   # the file itself is valid
    assert np.array_equal(decode(tmp_path / "i.png", tmp_path), arr)


def test_pnm_and_garbage(tmp_path):
    rng = np.random.RandomState(9)
    rgb = rng.randint(0, 256, size=(5, 6, 3), dtype=np.uint8)
    (tmp_path / "a.ppm").write_bytes(b"P6\n# comment\n6 5\n255\n" + rgb.tobytes())
    assert np.array_equal(decode(tmp_path / "a.ppm", tmp_path), rgb)
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    p = subprocess.run([TOOL, "decode", str(tmp_path / "bad.jpg"), str(tmp_path / "x.raw")], capture_output=True)
    assert p.returncode == 1
    (tmp_path / "junk.bin").write_bytes(os.urandom(64))
    assert subprocess.run([TOOL, "decode", str(tmp_path / "junk.bin"), str(tmp_path / "x.raw")], capture_output=True).returncode == 1


def test_malformed_inputs_are_rejected_not_crashed(tmp_path):
    """Truncated / corrupted JPEG and PNG files (the CLI sits behind an upload service in the reference's wrappers):
    the decoder may refuse them (exit 1) or decode what is there (exit 0), but never crash.  The ASan/UBSan version of
    this loop is tools/fuzz_codec.py."""
    import random
    rnd = random.Random(11)
    jpg = open(os.path.join(GOLDEN, "test.jpg"), "rb").read()
    png_path = tmp_path / "seed.png"
    encode(np.random.RandomState(2).randint(0, 256, size=(40, 56, 3), dtype=np.uint8), png_path, tmp_path)
    png = png_path.read_bytes()
    for k in range(40):
        src = jpg if k % 2 == 0 else png
        m = bytearray(src)
        if k % 4 < 2:
            m = m[: rnd.randrange(1, len(m))]
        else:
            for _ in range(rnd.randrange(1, 8)):
                m[rnd.randrange(0, min(len(m), 1024))] = rnd.randrange(256)
        f = tmp_path / "m.bin"
        f.write_bytes(bytes(m))
        p = subprocess.run([TOOL, "decode", str(f), str(tmp_path / "x.raw")], capture_output=True, timeout=60)
        assert p.returncode in (0, 1), (k, p.returncode, p.stderr[-300:])
