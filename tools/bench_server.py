#!/usr/bin/env python3
"""Throughput of the resident server under concurrent, unchanged CLI clients (what the Flask / worker wrappers generate):
N worker threads each run `bin/grayscale in out 1` in a loop with GRAYHIP_SERVER set.  Developer tool, GPU box only."""
import os, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
img = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "test.jpg")
tmp = tempfile.mkdtemp()
sock = os.path.join(tmp, "gh.sock")
# the server's own environment decides its PNG writer (GRAYHIP_PNG=fast: band-parallel; default: the reference's byte stream)
srv = subprocess.Popen([os.path.join(ROOT, "bin", "grayhipd"), sock], stdout=subprocess.PIPE, text=True)
srv.stdout.readline()
exe = os.path.join(ROOT, "bin", "grayscale")
def job(i, env):
    return subprocess.run([exe, img, os.path.join(tmp, f"o{i % 64}.png"), "1"], capture_output=True, env=env).returncode
try:
    for label, env in (("in-process", dict(os.environ)), ("via grayhipd", dict(os.environ, GRAYHIP_SERVER=sock))):
        # in-process callers each open the GPU: keep them few (the pool's process guard allows 6); served callers never touch it
        for workers in ((1, 4) if label == "in-process" else (1, 4, 16)):
            n = 8 * workers if label == "in-process" else 24 * workers
            t0 = time.perf_counter()
            with ThreadPoolExecutor(workers) as ex:
                rcs = list(ex.map(lambda i: job(i, env), range(n)))
            dt = time.perf_counter() - t0
            assert not any(rcs)
            print(f"{label:13s} {workers:2d} concurrent callers: {n/dt:7.1f} images/s  ({1e3*dt/n*workers:6.1f} ms per call)")
finally:
    srv.terminate(); srv.wait()
