#!/bin/bash
# round 2, GPU call H: chunked host shim (parity first), its PCIe-inclusive rate for a few chunk sizes
set -o pipefail
O=gpurun_out/r2h
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_shim or host_pointer" > $O/pytest_host.log 2>&1 || { tail -30 $O/pytest_host.log; exit 1; }
tail -2 $O/pytest_host.log
for mpx in 8 16 32 64; do echo "== GH_HOST_CHUNK_MPX=$mpx"; GH_HOST_CHUNK_MPX=$mpx python3 tools/bench_host_abi.py 2>&1 | grep -v amdgpu.ids | grep convert; done | tee $O/host_abi_chunks.txt
echo "== unchunked"; GH_HOST_PIPELINE_MIN_MB=100000 python3 tools/bench_host_abi.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_abi_chunks.txt
