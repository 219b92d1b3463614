"""The C-ABI shared library loads on a CPU-only box and exports exactly what include/*.h declare.
No compute call is made without a GPU -- except to prove that the product path FAILS LOUDLY there
(there is no CPU fallback behind the ABI)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, _gpu_available


def declared_symbols():
    names = set()
    text = open(os.path.join(ROOT, "include", "grayhip.h")).read()
    names.update(re.findall(r"GH_API\s+[\w\s\*]+?\b(\w+)\s*\(", text))
    for hdr in ("parallel_to_grayscale.h", "sobel.h"):
        t = open(os.path.join(ROOT, "include", hdr)).read()
        names.update(re.findall(r"^\s*void\s+(\w+)\s*\(", t, flags=re.M))
    return names


def test_header_declares_the_reference_prototypes():
    t = open(os.path.join(ROOT, "include", "parallel_to_grayscale.h")).read()
    assert "void convert_to_grayscale(unsigned char *data, int width, int height, int channels);" in t
    t = open(os.path.join(ROOT, "include", "sobel.h")).read()
    assert re.search(r"void sobel_edge\(const unsigned char \*src,\s*unsigned char \*dst,\s*int width,\s*int height\);", t)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.abi.load()
    names = declared_symbols()
    assert {"convert_to_grayscale", "sobel_edge", "gh_gray_rgb8_to_plane_dev", "gh_sobel_u8_rows_dev"} <= names
    assert names == set(pkg.abi.SIGNATURES), names ^ set(pkg.abi.SIGNATURES)
    for n in names:
        assert getattr(lib, n) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert names <= exported
    # nothing but the ABI (and toolchain-generated symbols) leaks out of the library
    leaked = {s for s in exported - names if not s.startswith(("_", "__hip"))}
    assert not leaked, leaked
    assert lib.gh_version().startswith(b"grayhip")


def test_c_caller_links_against_the_reference_headers(tmp_path, pkg):
    """A main.c-style caller compiles against include/*.h and links libgrayhip.so unchanged."""
    src = tmp_path / "caller.c"
    src.write_text(
        '#include "parallel_to_grayscale.h"\n#include "sobel.h"\n#include "grayhip.h"\n'
        "int main(int argc, char **argv) { if (argc > 99) { convert_to_grayscale(0, 0, 0, 3); sobel_edge(0, 0, 0, 0); }\n"
        "  return gh_version()[0] == 'g' ? 0 : 1; }\n")
    exe = tmp_path / "caller"
    libdir = os.path.dirname(pkg.abi.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT}/include", str(src), f"-L{libdir}", "-lgrayhip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True)
    assert subprocess.run([str(exe)]).returncode == 0


@pytest.mark.skipif(_gpu_available(), reason="only meaningful where no GPU is present")
def test_product_path_fails_loudly_without_gpu(pkg):
    img = np.zeros((8, 8, 3), dtype=np.uint8)
    with pytest.raises(pkg.GrayHipError) as ei:
        pkg.convert_to_grayscale(img, 8, 8, 3)
    assert "no HIP device" in str(ei.value) or "HIP" in str(ei.value)
    with pytest.raises(pkg.GrayHipError):
        pkg.sobel_edge(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8), 8, 8)
    # argument errors are reported without touching the GPU
    lib = pkg.abi.load()
    assert lib.gh_gray_inplace_dev(None, 16, 7, 0, None) == -2
    assert b"channels" in lib.gh_last_error()
    assert lib.gh_device_count() == 0
    # the multi-GPU layer and the resident server have no CPU mode either
    with pytest.raises(pkg.GrayHipError):
        pkg.sharding.NativeHaloComm(0, 1)
    p = subprocess.run([os.path.join(ROOT, "bin", "grayhipd"), "/tmp/never.sock"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr and not os.path.exists("/tmp/never.sock")
    # the RCCL dependency is optional for everything else: libgrayhip.so itself does not link librccl
    out = subprocess.run(["ldd", os.path.join(ROOT, "openmp-parallel-computing_amd", "lib", "libgrayhip.so")], capture_output=True, text=True).stdout
    assert "rccl" not in out


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package, the CLIs or the headers may reference it."""
    pkgdir = os.path.join(ROOT, "openmp-parallel-computing_amd")
    bad = []
    for base, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"pyoracle|pixel_oracle|liboracle|oracle/_ref|libref_", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(pkgdir, "lib", "libgrayhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libref" not in out


def test_historic_wrappers_refuse_frames_that_are_not_three_channel(pkg):
    """The old/ entry points read three longs per pixel and the pointer trees are built with 24-byte cells: any other trailing
    dimension is refused before anything reaches the library (no GPU needed to see that)."""
    api = pkg.hostapi
    for bad in (np.zeros((4, 5, 4), np.int64), np.zeros((4, 5), np.int64), np.zeros((4, 5, 1), np.int64)):
        with pytest.raises(pkg.GrayHipError):
            api.old_parallel_avg_pixel(bad)
        with pytest.raises(pkg.GrayHipError):
            api.old_parallel_to_grayscale(bad)
    with pytest.raises(pkg.GrayHipError):
        api.old_parallel_convolution(np.zeros((6, 7, 3), np.int64), [1] * 9, np.zeros((4, 5, 4), np.int64), 9)
    with pytest.raises(pkg.GrayHipError):
        api.old_parallel_convolution(np.zeros((6, 7, 4), np.int64), [1] * 9, np.zeros((4, 5, 3), np.int64), 9)


RELEASE_ENV_KNOBS = {"GH_SHARD_STEP_FORM", "GH_HALO_WAIT_MS", "GRAYHIP_MODE", "GRAYHIP_DEVICE", "GRAYHIP_VERBOSE"}


def _env_names(path):
    data = open(path, "rb").read()
    return {m.decode() for m in re.findall(rb"\b(?:GH|GRAYHIP)_[A-Z][A-Z0-9_]{2,}\b", data)} - {"GH_API", "GH_OK"}


def test_release_library_reads_only_the_documented_knobs(pkg):
    """`make lib` is the release build: the measurement lab's tuning knobs (GH_SOBEL_*, GH_FUSED_*, GH_CONV_*, ...) and the kernel variants
    they select are compiled out -- the only environment names in libgrayhip.so are the five INTEGRATION.md documents (the drop-in CLIs
    add GRAYHIP_SERVER and GRAYHIP_PNG, abi.py GRAYHIP_LIB).  The probe build (`make probes`) is where the lab lives."""
    names = _env_names(pkg.abi.LIB_PATH)
    assert names == RELEASE_ENV_KNOBS, sorted(names ^ RELEASE_ENV_KNOBS)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in sorted(RELEASE_ENV_KNOBS | {"GRAYHIP_SERVER", "GRAYHIP_PNG", "GRAYHIP_LIB"}):
        assert n in doc, f"{n} is not documented in INTEGRATION.md"
    if os.path.exists(pkg.abi.LAB_LIB_PATH):
        lab = _env_names(pkg.abi.LAB_LIB_PATH)
        assert len(lab) >= 40 and RELEASE_ENV_KNOBS <= lab
        assert os.path.getsize(pkg.abi.LIB_PATH) < 0.6 * os.path.getsize(pkg.abi.LAB_LIB_PATH)
        # the probe build exports the same C ABI (tests and tools swap the two freely)
        for name in pkg.abi.SIGNATURES:
            assert hasattr(ctypes.CDLL(pkg.abi.LAB_LIB_PATH), name), name
    for sym in ("k_sobel_dma4", "k_sobel_tiled", "k_wide_strip_probe", "k_gray_sobel_fused", "k_march_probe"):
        assert sym.encode() not in open(pkg.abi.LIB_PATH, "rb").read(), f"{sym} (a variant that lost its A/B) is in the release library"
