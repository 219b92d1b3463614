"""Drop-in proof at the SYMBOL boundary: the reference's own drivers (monolithic/src/main.c and main_with_sobel.c,
compiled in place by oracle/Makefile together with the stb headers the reference vendors) linked against
libgrayhip.so instead of the reference's kernels.  The binaries are built in the build container
(oracle/_ref/refmain*_on_grayhip, prebuilt files travel to the GPU box); the reference sources themselves do not."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, ROOT

GRAY = os.path.join(ROOT, "oracle", "_ref", "refmain_on_grayhip")
SOBEL = os.path.join(ROOT, "oracle", "_ref", "refmain_sobel_on_grayhip")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.access(GRAY, os.X_OK) and os.access(SOBEL, os.X_OK)),
                                 reason="oracle/_ref drop-in binaries are built only where /root/reference exists")]


def sha_png(path):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(path))).data).hexdigest()


def test_reference_main_c_on_libgrayhip(tmp_path):
    full = json.load(open(os.path.join(GOLDEN, "test_jpg_full.json")))
    out = tmp_path / "o.png"
    p = subprocess.run([GRAY, os.path.join(GOLDEN, "test.jpg"), str(out), "1"], capture_output=True, text=True)
    assert p.returncode == 0 and p.stderr == "" and p.stdout.startswith("Compute kernel ×1: ")
    assert sha_png(out) == full["gray_ieee_rgb_sha256"]          # == the reference's golden images/test_gray.png
    p = subprocess.run([GRAY, os.path.join(GOLDEN, "test.jpg"), str(out), "2"], capture_output=True, text=True)
    assert p.returncode == 0 and sha_png(out) == full["cli_gray_p2_rgb_sha256"]


def test_reference_main_with_sobel_c_on_libgrayhip(tmp_path):
    full = json.load(open(os.path.join(GOLDEN, "test_jpg_full.json")))
    out = tmp_path / "o.png"
    for passes in (1, 2):
        p = subprocess.run([SOBEL, os.path.join(GOLDEN, "test.jpg"), str(out), str(passes)], capture_output=True, text=True)
        assert p.returncode == 0 and p.stderr == ""
        assert sha_png(out) == full[f"cli_sobel_p{passes}_rgb_sha256"], passes
