"""The drop-in CLI contract of the reference's drivers (monolithic/src/main.c, main_with_sobel.c):
    bin/grayscale <input_img> <output_img.png> [passes]
same argv handling, stdout line, stderr texts and exit codes; what the reference's bench script and the Flask /
RabbitMQ wrappers rely on.  Error paths run anywhere; full runs need the GPU."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, ROOT

GRAY = os.path.join(ROOT, "bin", "grayscale")
SOBEL = os.path.join(ROOT, "bin", "grayscale_sobel")
SERVER = os.path.join(ROOT, "bin", "grayhipd")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.access(GRAY, os.X_OK) and os.access(SOBEL, os.X_OK) and os.access(SERVER, os.X_OK)):
        subprocess.run(["make", "-C", ROOT, "cli"], check=True, capture_output=True)


def run(exe, *args, env=None):
    e = dict(os.environ)
    e.pop("GRAYHIP_VERBOSE", None)
    if env:
        e.update(env)
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=e)


@pytest.mark.parametrize("exe", [GRAY, SOBEL])
def test_usage_and_load_errors(exe, tmp_path):
    p = run(exe)
    assert p.returncode == 1 and p.stdout == ""
    assert p.stderr == f"Uso: {exe} <input_img> <output_img.png> [passaggi_kernel]\n"        # main.c:16
    p = run(exe, "only_one_arg")
    assert p.returncode == 1 and p.stderr.startswith("Uso: ")
    missing = tmp_path / "nope.png"
    p = run(exe, missing, tmp_path / "o.png")
    assert p.returncode == 1 and p.stdout == ""
    if exe == GRAY:
        assert p.stderr == "Errore caricando immagine\n"                                     # main.c:23
    else:
        assert p.stderr == f'Errore caricando immagine "{missing}"\n'                        # main_with_sobel.c:30-31
    junk = tmp_path / "junk.jpg"
    junk.write_bytes(b"definitely not an image")
    assert run(exe, junk, tmp_path / "o.png").returncode == 1


@pytest.mark.gpu
def test_gray_cli_reproduces_the_reference_golden_file(tmp_path):
    """bin/grayscale images/test.jpg out.png -> the pixels of the reference's committed images/test_gray.png."""
    full = json.load(open(os.path.join(GOLDEN, "test_jpg_full.json")))
    out = tmp_path / "out.png"
    p = run(GRAY, os.path.join(GOLDEN, "test.jpg"), out)
    assert p.returncode == 0 and p.stderr == ""                       # stderr must stay empty (bench script parses it)
    assert re.fullmatch(r"Compute kernel ×1: \d+\.\d{4} s\n", p.stdout)
    img = np.asarray(Image.open(out))
    assert img.shape == (1080, 1920, 3)
    assert hashlib.sha256(np.ascontiguousarray(img).data).hexdigest() == full["gray_ieee_rgb_sha256"]
    # ... and the FILE is the reference's committed images/test_gray.png, byte for byte
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["test_jpg"]
    assert hashlib.md5(out.read_bytes()).hexdigest() == ka["golden_png_md5"]
    p = run(GRAY, os.path.join(GOLDEN, "test.jpg"), out, 2)
    assert p.returncode == 0 and p.stdout.startswith("Compute kernel ×2: ")
    assert hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(out))).data).hexdigest() == full["cli_gray_p2_rgb_sha256"]
    assert hashlib.md5(out.read_bytes()).hexdigest() == ka["cli_gray_ieee_p2_png_md5"]
    p = run(GRAY, os.path.join(GOLDEN, "test.jpg"), out, env={"GRAYHIP_MODE": "fma"})
    assert hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(out))).data).hexdigest() == full["gray_fma_rgb_sha256"]
    assert hashlib.md5(out.read_bytes()).hexdigest() == ka["cli_gray_fma_p1_png_md5"]
    # GRAYHIP_PNG=fast: the band-parallel encoder -- another byte stream, the same pixels
    p = run(GRAY, os.path.join(GOLDEN, "test.jpg"), out, env={"GRAYHIP_PNG": "fast"})
    assert p.returncode == 0 and hashlib.md5(out.read_bytes()).hexdigest() != ka["golden_png_md5"]
    assert hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(out))).data).hexdigest() == full["gray_ieee_rgb_sha256"]


@pytest.mark.gpu
def test_sobel_cli_matches_reference_cli(tmp_path):
    full = json.load(open(os.path.join(GOLDEN, "test_jpg_full.json")))
    out = tmp_path / "out.png"
    for passes in (1, 2):
        p = run(SOBEL, os.path.join(GOLDEN, "test.jpg"), out, passes)
        assert p.returncode == 0 and p.stderr == ""
        assert re.fullmatch(rf"Compute kernel \(grayscale \+ sobel\) ×{passes}: \d+\.\d{{4}} s\n", p.stdout)
        got = hashlib.sha256(np.ascontiguousarray(np.asarray(Image.open(out))).data).hexdigest()
        assert got == full[f"cli_sobel_p{passes}_rgb_sha256"], passes
        ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["test_jpg"]
        assert hashlib.md5(out.read_bytes()).hexdigest() == ka[f"cli_sobel_ieee_p{passes}_png_md5"], passes     # file-level parity


@pytest.mark.gpu
def test_passes_parsing_rgba_and_save_failures(tmp_path, jpg_crop, cli_crop):
    rgb = jpg_crop["rgb"]
    src = tmp_path / "crop.png"
    Image.fromarray(rgb).save(src)
    out = tmp_path / "o.png"
    for arg, expect in (("0", 1), ("abc", 1), ("-5", 1), ("3", 3)):              # atoi, <1 -> 1  (main.c:27-28)
        p = run(GRAY, src, out, arg)
        assert p.returncode == 0 and p.stdout.startswith(f"Compute kernel ×{expect}: ")
    assert np.array_equal(np.asarray(Image.open(out))[..., 0], cli_crop["gray_rgb_p2_c0"])   # passes=3 == passes=2
    rgba = np.dstack([rgb, cli_crop["rgba_alpha"]])
    srca = tmp_path / "crop_rgba.png"
    Image.fromarray(rgba, "RGBA").save(srca)
    assert run(SOBEL, srca, out, 2).returncode == 0
    got = np.asarray(Image.open(out))
    assert got.shape == rgba.shape                                               # same channel count as the input
    assert np.array_equal(got[..., 0], cli_crop["sobel_rgba_p2_c0"]) and np.array_equal(got[..., 3], rgba[..., 3])
    bad = tmp_path / "no_such_dir" / "o.png"
    p = run(GRAY, src, bad)
    assert p.returncode == 1 and p.stderr == "Errore nel salvataggio\n" and p.stdout.startswith("Compute kernel")
    p = run(SOBEL, src, bad)
    assert p.returncode == 0 and p.stderr == f'Errore nel salvataggio di "{bad}"\n'       # exit code stays 0 (main_with_sobel.c:83-91)


@pytest.mark.gpu
def test_flask_style_caller(tmp_path, jpg_crop):
    """What microservices/grayscale/app.py:24-38 does: exec [binary, in, out(, passes)], needs rc 0 and out.png."""
    src = tmp_path / "upload.png"
    Image.fromarray(jpg_crop["rgb"]).save(src)
    out = tmp_path / "out.png"
    env = dict(os.environ, OMP_NUM_THREADS="4")
    result = subprocess.run([GRAY, str(src), str(out), "2"], capture_output=True, env=env)
    assert result.returncode == 0 and out.exists()
    assert np.array_equal(np.asarray(Image.open(out))[..., 0], np.asarray(Image.open(out))[..., 2])


# ---- the resident server (bin/grayhipd) and the CLIs as its clients ----------------------------------------

def _field(s: bytes) -> bytes:
    return str(len(s)).encode() + b"\n" + s


def test_client_speaks_the_wire_protocol_and_falls_back(tmp_path):
    """No GPU needed: a stand-in server checks what the CLI sends and feeds it a canned reply; with no server at the
    socket path the CLI does the work itself (and, on a box without a GPU, fails loudly as before)."""
    import socket
    import threading
    path = str(tmp_path / "s.sock")
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(path)
    srv.listen(1)
    seen = {}

    def serve():
        c, _ = srv.accept()
        f = c.makefile("rb")
        seen["head"] = f.readline()
        fields = []
        for _ in range(4):
            n = int(f.readline())
            fields.append(f.read(n))
        seen["fields"] = fields
        c.sendall(b"7\n" + _field("Compute kernel ×3: 0.0001 s\n".encode()) + _field(b"a warning\n"))
        c.close()

    t = threading.Thread(target=serve)
    t.start()
    os.makedirs(tmp_path / "wd", exist_ok=True)
    e = dict(os.environ, GRAYHIP_SERVER=path, GRAYHIP_MODE="fma")
    p = subprocess.run([GRAY, "in.jpg", "/abs/out.png", "3"], capture_output=True, text=True, env=e, cwd=tmp_path / "wd")
    t.join()
    srv.close()
    assert seen["head"] == b"GRAYHIP1 gray 4\n"
    assert seen["fields"] == [str(tmp_path / "wd" / "in.jpg").encode(), b"/abs/out.png", b"3", b"fma"]     # paths made absolute
    assert (p.returncode, p.stdout, p.stderr) == (7, "Compute kernel ×3: 0.0001 s\n", "a warning\n")
    # nobody listening: in-process path, same contract as without GRAYHIP_SERVER
    p = subprocess.run([GRAY, tmp_path / "missing.png", tmp_path / "o.png"], capture_output=True, text=True,
                       env=dict(os.environ, GRAYHIP_SERVER=str(tmp_path / "nobody.sock")))
    assert p.returncode == 1 and p.stderr == "Errore caricando immagine\n"


@pytest.mark.gpu
def test_resident_server_serves_the_unchanged_cli(tmp_path, jpg_crop):
    """bin/grayhipd + GRAYHIP_SERVER: same stdout / stderr / exit codes and the same output pixels as the in-process CLI,
    also under concurrent clients and with the arithmetic mode of each client."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    sock = str(tmp_path / "gh.sock")
    srv = subprocess.Popen([SERVER, sock], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        assert srv.stdout.readline().startswith("grayhipd: serving on ")
        src = tmp_path / "in.png"
        Image.fromarray(jpg_crop["rgb"]).save(src)
        env = {"GRAYHIP_SERVER": sock}
        for exe, line in ((GRAY, r"^Compute kernel ×2: \d+\.\d{4} s\n$"), (SOBEL, r"^Compute kernel \(grayscale \+ sobel\) ×2: \d+\.\d{4} s\n$")):
            local, served = tmp_path / "local.png", tmp_path / "served.png"
            a = run(exe, src, local, 2)
            b = run(exe, src, served, 2, env=env)
            assert a.returncode == b.returncode == 0 and b.stderr == "" and re.match(line, b.stdout), (b.stdout, b.stderr)
            assert np.array_equal(np.asarray(Image.open(local)), np.asarray(Image.open(served)))
        # per-client mode
        outs = {}
        for mode in ("ieee", "fma"):
            o = tmp_path / f"{mode}.png"
            assert run(GRAY, src, o, env={**env, "GRAYHIP_MODE": mode}).returncode == 0
            outs[mode] = np.asarray(Image.open(o))
            lo = tmp_path / f"{mode}_local.png"
            assert run(GRAY, src, lo, env={"GRAYHIP_MODE": mode}).returncode == 0
            assert np.array_equal(outs[mode], np.asarray(Image.open(lo)))
        # error paths keep their texts and codes
        p = run(GRAY, tmp_path / "nope.png", tmp_path / "o.png", env=env)
        assert (p.returncode, p.stdout, p.stderr) == (1, "", "Errore caricando immagine\n")
        p = run(SOBEL, src, tmp_path / "no_such_dir" / "o.png", env=env)
        assert p.returncode == 0 and p.stderr.startswith("Errore nel salvataggio di")
        # concurrent clients
        def one(i):
            o = tmp_path / f"c{i}.png"
            r = run(SOBEL if i % 2 else GRAY, src, o, 1, env=env)
            return r.returncode, o
        with ThreadPoolExecutor(8) as ex:
            res = list(ex.map(one, range(16)))
        ref = {0: np.asarray(Image.open(res[0][1])), 1: np.asarray(Image.open(res[1][1]))}
        for i, (rc, o) in enumerate(res):
            assert rc == 0 and np.array_equal(np.asarray(Image.open(o)), ref[i % 2])
        # what it is for: per-call wall time without HIP start-up
        t0 = time.perf_counter(); run(GRAY, src, tmp_path / "t.png", env=env); served_s = time.perf_counter() - t0
        t0 = time.perf_counter(); run(GRAY, src, tmp_path / "t.png"); local_s = time.perf_counter() - t0
        print(f"served {served_s*1e3:.0f} ms vs in-process {local_s*1e3:.0f} ms")
    finally:
        srv.terminate()
        srv.wait(timeout=20)
    assert not os.path.exists(sock)          # the server removes its socket on SIGTERM
