#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference (build container only).

What it does (nothing from /root/reference is copied as source; only pixel DATA is kept):
  1. decodes /root/reference/images/test.jpg with the reference's own vendored stb_image.h
     (oracle/_ref/stb_decode, built by oracle/Makefile from oracle/tools/stb_decode.c) and checks the
     SHA-256 prefix recorded in SURVEY.md section 8c;
  2. runs the COMPILED REFERENCE (oracle/_ref/libref_ieee.so, libref_fma_native.so) on the full
     frame: convert_to_grayscale -> plane -> sobel_edge;
  3. checks that the IEEE gray equals the reference's committed golden images/test_gray.png;
  4. keeps a 512x320 crop (x=704, y=380) of: decoded RGB, golden PNG, IEEE/FMA gray, IEEE/FMA edge
     (edge crop = crop of the full-frame edge, so only its interior is comparable to Sobel(crop));
  5. writes the crop as a PNG (RGB and RGBA), runs the reference CLIs (oracle/_ref/grayscale_ieee,
     grayscale_sobel_ieee) on it with passes 1,2,3 and keeps channel 0 of their decoded outputs
     (R==G==B and alpha-unchanged are asserted here).

Output: tests/golden/test_jpg_crop.npz, tests/golden/cli_crop.npz   (loaded with allow_pickle=False)
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

REF = "/root/reference"
X0, Y0, CW, CH = 704, 380, 512, 320


def main() -> None:
    O.build(force=True)
    refdir = O.REF_DIR
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "test.raw")
        out = subprocess.run([os.path.join(refdir, "stb_decode"), f"{REF}/images/test.jpg", raw],
                             check=True, capture_output=True, text=True).stdout.split()
        w, h, c = map(int, out)
        rgb = np.fromfile(raw, dtype=np.uint8).reshape(h, w, c)
        digest = hashlib.sha256(rgb.data).hexdigest()
        assert digest.startswith("3d22bc091f9d3039"), digest
        assert (w, h, c) == (1920, 1080, 3)

        golden = np.asarray(Image.open(f"{REF}/images/test_gray.png"))
        assert golden.shape == rgb.shape

        ieee, fma = O.RefLib("ieee"), O.RefLib("fma")
        g_ieee = ieee.convert_to_grayscale(rgb)
        g_fma = fma.convert_to_grayscale(rgb)
        assert np.array_equal(g_ieee, golden), "IEEE build must reproduce images/test_gray.png"
        ndiff = int((g_ieee[..., 0] != g_fma[..., 0]).sum())
        assert ndiff == 145, ndiff
        e_ieee = ieee.sobel_edge(g_ieee[..., 0])
        e_fma = ieee.sobel_edge(g_fma[..., 0])

        # our restatement must agree with the compiled reference on the full frame
        assert np.array_equal(O.gray_inplace(rgb, O.MODE_IEEE), g_ieee)
        assert np.array_equal(O.gray_inplace(rgb, O.MODE_FMA), g_fma)
        assert np.array_equal(O.sobel(g_ieee[..., 0]), e_ieee)

        sl = (slice(Y0, Y0 + CH), slice(X0, X0 + CW))
        crop = np.ascontiguousarray(rgb[sl])
        np.savez_compressed(
            os.path.join(ROOT, "tests/golden/test_jpg_crop.npz"),
            origin=np.array([X0, Y0]), rgb=crop, golden_png=np.ascontiguousarray(golden[sl]),
            gray_ieee=np.ascontiguousarray(g_ieee[..., 0][sl]), gray_fma=np.ascontiguousarray(g_fma[..., 0][sl]),
            edge_ieee=np.ascontiguousarray(e_ieee[sl]), edge_fma=np.ascontiguousarray(e_fma[sl]),
            full_gray_ieee_sha256=np.frombuffer(hashlib.sha256(g_ieee.data).digest(), dtype=np.uint8),
            full_edge_ieee_sha256=np.frombuffer(hashlib.sha256(e_ieee.data).digest(), dtype=np.uint8),
        )

        # reference CLIs on the crop (PNG in, PNG out; both lossless, so PIL decode == stb decode)
        png_in = os.path.join(tmp, "crop.png")
        Image.fromarray(crop).save(png_in)
        rgba = np.dstack([crop, O.frame_rand(CW, CH, 1, seed=7)[..., 0]])
        png_rgba = os.path.join(tmp, "crop_rgba.png")
        Image.fromarray(rgba, "RGBA").save(png_rgba)
        res = {"rgb": crop, "rgba": rgba}
        for exe, tag in (("grayscale_ieee", "gray"), ("grayscale_sobel_ieee", "sobel")):
            for passes in (1, 2, 3):
                for src, stag in ((png_in, "rgb"), (png_rgba, "rgba")):
                    dst = os.path.join(tmp, "o.png")
                    p = subprocess.run([os.path.join(refdir, exe), src, dst, str(passes)],
                                       check=True, capture_output=True, text=True)
                    assert p.stderr == ""
                    img = np.asarray(Image.open(dst))
                    # R == G == B on output and alpha untouched: checked here, so only channel 0 is stored
                    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
                    if stag == "rgba":
                        assert np.array_equal(img[..., 3], rgba[..., 3])
                    res[f"{tag}_{stag}_p{passes}_c0"] = np.ascontiguousarray(img[..., 0])
                    res[f"{tag}_{stag}_p{passes}_stdout"] = np.frombuffer(p.stdout.encode(), dtype=np.uint8)
        assert np.array_equal(res["gray_rgb_p2_c0"], res["gray_rgb_p3_c0"])   # output stabilises from pass 2
        for k in ("gray_rgb_p3_c0", "gray_rgba_p3_c0", "sobel_rgba_p3_c0", "gray_rgba_p2_c0"):
            del res[k]
        res["rgba_alpha"] = np.ascontiguousarray(rgba[..., 3])
        del res["rgb"], res["rgba"]      # rgb == test_jpg_crop.npz:rgb ; rgba = rgb + rgba_alpha
        np.savez_compressed(os.path.join(ROOT, "tests/golden/cli_crop.npz"), **res)
        # full-frame digests for the CLI file-level tests (tests/golden/test.jpg is a copy of the reference's images/test.jpg)
        import json
        full = {"decoded_rgb_sha256": digest,
                "gray_ieee_rgb_sha256": hashlib.sha256(g_ieee.data).hexdigest(),      # == decoded images/test_gray.png
                "gray_fma_rgb_sha256": hashlib.sha256(g_fma.data).hexdigest()}
        for exe, tag in (("grayscale_ieee", "cli_gray"), ("grayscale_sobel_ieee", "cli_sobel")):
            for passes in (1, 2):
                dst = os.path.join(tmp, "full.png")
                subprocess.run([os.path.join(refdir, exe), f"{REF}/images/test.jpg", dst, str(passes)], check=True, capture_output=True)
                full[f"{tag}_p{passes}_rgb_sha256"] = hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(dst))).data).hexdigest()
        assert full["cli_gray_p1_rgb_sha256"] == full["gray_ieee_rgb_sha256"]
        for name in ("half_of_a_mega_photo.jpg", "more_than_one_mega_photo.jpg"):
            raw2 = os.path.join(tmp, "o.raw")
            subprocess.run([os.path.join(refdir, "stb_decode"), f"{REF}/images/{name}", raw2], check=True, capture_output=True)
            full[f"decoded_{name}_sha256"] = hashlib.sha256(open(raw2, "rb").read()).hexdigest()
        with open(os.path.join(ROOT, "tests/golden/test_jpg_full.json"), "w") as f:
            json.dump(full, f, indent=1)
        b = res["sobel_rgb_p1_c0"]
        print("fixtures written; sobel p1 border all zero:",
              bool((b[0] == 0).all() and (b[-1] == 0).all() and (b[:, 0] == 0).all() and (b[:, -1] == 0).all()))


if __name__ == "__main__":
    main()
