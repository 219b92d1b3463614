"""Dynamic counterpart of tests/test_isa_vmcnt.py: a bounded, seeded run of tools/stress_parity.py -- random shapes (any
width), channel counts, modes, band heights, ring depths and dispatch orders of the LDS-DMA kernels, every result compared
with the oracle, while a second stream keeps 1 GiB of copy traffic in flight (memory-latency noise: a wrong vmcnt count
shows up only when a row arrives late)."""
import importlib.util
import os

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_lds_dma_kernels_under_latency_noise():
    # (the tool switches to the probe build itself: its knob sets select ring depths, band heights and dispatch orders)
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(ROOT, "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.run(float(os.environ.get("GH_STRESS_SECONDS", "40")), seed=20261005)
    assert n >= 200, f"only {n} results in the time budget"
    assert bad == 0, f"{bad} of {n} results differ from the oracle"
