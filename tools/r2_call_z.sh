#!/bin/bash
# round 2, GPU call Z: chunked sobel_edge(host pointers): parity, then PCIe-inclusive timing against the single-copy form
set -o pipefail
O=gpurun_out/r2z
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "chunked or host_shim" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python3 tools/bench_host_abi.py > $O/host_abi_chunked.txt 2>&1 || { tail $O/host_abi_chunked.txt; exit 1; }
GH_HOST_PIPELINE_MIN_MB=1000000 timeout -k 10 300 python3 tools/bench_host_abi.py > $O/host_abi_single.txt 2>&1 || { tail $O/host_abi_single.txt; exit 1; }
echo "== chunked"; grep -v amdgpu $O/host_abi_chunked.txt; echo "== single copy"; grep -v amdgpu $O/host_abi_single.txt
