"""The host-side launch rules of the strip kernels (stream_core.h: band_start_phases, prefer_short_bands, wg_balance,
shifted_band_overflows) against a Python restatement and the cases DESIGN.md quotes.  The helper program is host code built with hipcc;
no GPU needed."""
import math
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def phases(stride):
    w = 1 << 20
    r = stride % w
    return 1 if r == 0 else w // math.gcd(r, w)


def prefer_short(pitch, tall, short):
    pt, ps = phases(tall * pitch), phases(short * pitch)
    return 1 < pt < 16 and (ps == 1 or ps > pt)


def balance(nb, cu=256):
    rounds = -(-nb // cu)
    return nb / (rounds * cu)


def overflows(groups, rows, r, cu=256):
    if rows % r == 0:
        return False
    n_full = groups * (rows // r)
    n_all = n_full + groups
    lf, la = -(-n_full // cu), -(-n_all // cu)
    return la > lf and lf < 8


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    if not os.path.exists(HIPCC) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    exe = tmp_path_factory.mktemp("rules") / "launch_rules_dump"
    src = os.path.join(ROOT, "tests", "native", "launch_rules_dump.hip")
    inc = os.path.join(ROOT, "openmp-parallel-computing_amd", "csrc")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + inc, src, "-o", str(exe)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    return [line.split() for line in out.splitlines()]


def test_helpers_equal_their_restatement(dump):
    seen = set()
    for rec in dump:
        kind = rec[0]
        seen.add(kind)
        if kind == "phases":
            assert int(rec[3]) == phases(int(rec[2]) * int(rec[1])), rec
        elif kind == "prefer":
            assert bool(int(rec[4])) == prefer_short(int(rec[1]), int(rec[2]), int(rec[3])), rec
        elif kind == "balance":
            assert abs(float(rec[2]) - balance(int(rec[1]))) < 1e-5, rec
        elif kind == "overflow":
            assert bool(int(rec[4])) == overflows(int(rec[1]), int(rec[2]), int(rec[3])), rec
    assert seen >= {"phases", "prefer", "balance", "overflow", "rows_fused", "rows_sobel"}


def fused_rows(gx, rows, pitch, cu, beside_exchange=False):
    nb64, nb32 = gx * -(-rows // 64), gx * -(-rows // 32)
    if beside_exchange and nb64 < cu and cu <= nb32 < 2 * cu and balance(nb32, cu) >= 0.97:
        return 32
    rd = 64 if (nb64 >= 2 * cu or (nb64 >= cu and balance(nb64, cu) >= 0.97)) else (32 if (nb32 >= 2 * cu and balance(nb32, cu) >= 0.9) else 16)
    if rd == 64 and prefer_short(pitch, 64, 32) and nb32 >= cu:
        rd = 32
    if rd == 64 and prefer_short(pitch, 64, 16):
        rd = 16
    if rd == 32 and prefer_short(pitch, 32, 16):
        rd = 16
    return rd


def sobel_rows(gx, rows, pitch, cu):
    nb64, nb32 = gx * -(-rows // 64), gx * -(-rows // 32)
    r = 64 if nb64 >= 2 * cu else (32 if nb32 >= 2 * cu else 16)
    if r == 64 and prefer_short(pitch, 64, 16):
        r = 16
    if r == 32 and rows >= 16 and not overflows(gx, rows, 16, cu):
        r = 16
    return r


def test_band_heights_follow_the_device_cu_count(dump):
    """The launchers size their plans for the CU count the device reports (gh_init), not for a constant: the rules are pure functions
    of (geometry, CUs).  Checked against a restatement for 256 CUs (MI355X, SPX), 32 (a CPX partition) and 304; and the point of it:
    a partition with an eighth of the CUs gives a frame with an eighth of the rows the plan the whole chip gives the whole frame."""
    got = {(r[0], int(r[1]), int(r[2]), int(r[3])): int(r[4]) for r in dump if r[0] in ("rows_fused", "rows_sobel", "rows_fused_shard")}
    assert len(got) >= 90
    for (kind, cu, w, h), val in got.items():
        gx = -(-w // 4096)
        want = sobel_rows(gx, h, w, cu) if kind == "rows_sobel" else fused_rows(gx, h, 3 * w, cu, kind == "rows_fused_shard")
        assert val == want, (kind, cu, w, h, val, want)
    # the shard kernel of a sharded step (an exchange runs beside it): 32-row bands where they give every CU exactly one workgroup
    assert got[("rows_fused_shard", 256, 16384, 2048)] == 32 and got[("rows_fused", 256, 16384, 2048)] == 16
    assert got[("rows_fused_shard", 256, 16384, 4096)] == got[("rows_fused", 256, 16384, 4096)] == 64
    assert got[("rows_fused_shard", 256, 16384, 8192)] == 64 and got[("rows_fused_shard", 32, 16384, 512)] == got[("rows_fused_shard", 256, 16384, 4096)]
    assert got[("rows_fused", 256, 16384, 16384)] == 64 and got[("rows_fused", 256, 16384, 2048)] == 16        # the headline and the 8-GPU shard
    assert got[("rows_fused", 32, 16384, 2048)] == got[("rows_fused", 256, 16384, 16384)]                      # 1/8 of the CUs, 1/8 of the rows
    assert got[("rows_sobel", 32, 16384, 2048)] == got[("rows_sobel", 256, 16384, 16384)] == 64
    assert got[("rows_fused", 32, 16384, 512)] == got[("rows_fused", 256, 16384, 4096)]


def test_the_cases_design_md_quotes():
    # plane Sobel (pitch = width): 64-row bands alias on 12288 / 10240 / 20480 / 24576-wide planes, not on 16384 / 32768 / 16368 / 16370
    assert [phases(64 * w) for w in (12288, 24576, 20480, 10240)] == [4, 2, 4, 8]
    assert phases(64 * 16384) == 1 and phases(64 * 32768) == 1 and phases(64 * 16368) == 1024 and phases(16 * 16384) == 4
    for w in (12288, 10240, 20480, 24576):
        assert prefer_short(w, 64, 16), w
    for w in (16384, 32768, 16368, 16370, 15360):
        assert not prefer_short(w, 64, 16), w
    # fused kernel (pitch = 3 x width): 10240 wide -> 32-row bands, 12288 wide -> 16
    assert prefer_short(3 * 10240, 64, 32) and not prefer_short(3 * 10240, 32, 16)
    assert prefer_short(3 * 12288, 64, 16) and prefer_short(3 * 12288, 32, 16)
    assert not prefer_short(3 * 16384, 64, 32) and not prefer_short(3 * 16384, 64, 16)
    # balance: 384 workgroups on 256 CUs = 0.75, 480 = 0.94, 256 and 512 = 1
    assert balance(384) == 0.75 and round(balance(480), 2) == 0.94 and balance(256) == 1.0 and balance(512) == 1.0
    # a shifted last band must not add a layer of workgroups: 16384 x 8200 with two strips (258 on 256) does, 7680 x 4320 (2 x 270) does not
    assert overflows(2, 8200, 64) and overflows(1, 4100, 16) and not overflows(2, 4320, 16) and not overflows(1, 1080, 16)


def test_xcd_interleaved_orders_only_on_eight_xcds(dump):
    """band_of_block's orders 4 / 6 / 8 deal workgroups to "XCD b % 8"; on a device with another XCD count (a CPX partition has one)
    the request falls back to the band-major order 0 and the grid is not padded to multiples of 8 / 16 bands."""
    recs = [(int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[6])) for r in dump if r[0] == "order"]
    assert len(recs) >= 90
    for xcds, bands, gx, req, granted, grid in recs:
        m_req, m = req % 10, granted % 10
        if xcds != 8:
            assert m in (0, 2), (xcds, req, granted)
            assert grid == bands * gx, (xcds, bands, gx, req, grid)
        elif m_req in (6, 8) and bands >= 1:
            assert m == m_req
            pad = 8 if m == 6 else 16
            assert grid == -(-bands // pad) * pad * gx
        if m_req == 2:
            assert m == 2
    # the headline plan on the whole chip and on one XCD of it
    assert (8, 256, 4, 6, 6, 1024) in recs and (1, 32, 4, 6, 0, 128) in recs
