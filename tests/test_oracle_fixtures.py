"""The oracle against the reference's own golden image (images/test_gray.png, via the committed crop
in tests/golden/) and against outputs of the reference CLIs recorded by tests/golden/make_fixtures.py;
plus, in the build container only, against the reference sources compiled in place (oracle/_ref)."""
import os

import numpy as np
import pytest

HAVE_REFERENCE_TREE = os.path.isdir("/root/reference/monolithic/src")


def test_golden_png_crop_is_ieee_gray(oracle, jpg_crop):
    rgb = jpg_crop["rgb"]
    golden = jpg_crop["golden_png"]            # crop of the reference's committed images/test_gray.png
    got = oracle.gray_inplace(rgb, oracle.MODE_IEEE)
    assert np.array_equal(got, golden)
    assert np.array_equal(got[..., 0], jpg_crop["gray_ieee"])
    assert np.array_equal(oracle.gray_to_plane(rgb, oracle.MODE_FMA), jpg_crop["gray_fma"])
    assert (jpg_crop["gray_ieee"] != jpg_crop["gray_fma"]).sum() > 0   # the two modes really differ on this crop


def test_edge_crop_interior(oracle, jpg_crop):
    # the stored edge is a crop of the full-frame edge: only the crop's interior is comparable
    for mode in ("ieee", "fma"):
        edge = oracle.sobel(jpg_crop[f"gray_{mode}"])
        assert np.array_equal(edge[1:-1, 1:-1], jpg_crop[f"edge_{mode}"][1:-1, 1:-1])


def test_reference_cli_outputs(oracle, jpg_crop, cli_crop):
    rgb = jpg_crop["rgb"]
    rgba = np.dstack([rgb, cli_crop["rgba_alpha"]])
    # grayscale CLI (main.c): passes x convert_to_grayscale
    assert np.array_equal(oracle.gray_passes(rgb, 1)[..., 0], cli_crop["gray_rgb_p1_c0"])
    p2 = oracle.gray_passes(rgb, 2)
    assert np.array_equal(p2[..., 0], cli_crop["gray_rgb_p2_c0"])
    assert np.array_equal(oracle.gray_passes(rgb, 3), p2)          # stabilises from pass 2
    out = oracle.gray_passes(rgba, 1)
    assert np.array_equal(out[..., 0], cli_crop["gray_rgba_p1_c0"]) and np.array_equal(out[..., 3], rgba[..., 3])
    # grayscale+sobel CLI (main_with_sobel.c)
    for passes in (1, 2, 3):
        img, _, _ = oracle.pipeline(rgb, passes)
        assert np.array_equal(img[..., 0], cli_crop[f"sobel_rgb_p{passes}_c0"]), passes
        assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
    for passes in (1, 2):
        img, _, _ = oracle.pipeline(rgba, passes)
        assert np.array_equal(img[..., 0], cli_crop[f"sobel_rgba_p{passes}_c0"]), passes
        assert np.array_equal(img[..., 3], rgba[..., 3])


def test_reference_cli_stdout_format(cli_crop):
    assert bytes(cli_crop["gray_rgb_p2_stdout"]).decode().startswith("Compute kernel ×2: ")
    assert bytes(cli_crop["sobel_rgb_p3_stdout"]).decode().startswith("Compute kernel (grayscale + sobel) ×3: ")
    assert bytes(cli_crop["gray_rgb_p1_stdout"]).decode().endswith(" s\n")


@pytest.mark.skipif(not HAVE_REFERENCE_TREE, reason="reference tree only exists in the build container")
@pytest.mark.parametrize("which,mode", [("ieee", 0), ("fma", 1)])
def test_against_compiled_reference(oracle, which, mode):
    if not oracle.have_ref(which):
        oracle.build(force=True)
    ref = oracle.RefLib(which)
    for w, h, ch, seed in ((640, 360, 3, 11), (333, 77, 3, 12), (256, 256, 4, 13), (17, 5, 3, 14)):
        img = oracle.frame_rand(w, h, ch, seed=seed)
        want = ref.convert_to_grayscale(img)
        assert np.array_equal(oracle.gray_inplace(img, mode), want)
        plane = np.ascontiguousarray(want[..., 0])
        assert np.array_equal(oracle.sobel(plane), ref.sobel_edge(plane))
    cube = oracle.frame_cube()
    assert np.array_equal(oracle.gray_inplace(cube, mode), ref.convert_to_grayscale(cube))
