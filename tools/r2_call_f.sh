#!/bin/bash
# round 2, GPU call F: CLI wall times (both PNG modes), server throughput, final default bench + 2-rank rehearsals
set -o pipefail
O=gpurun_out/r2f
mkdir -p $O
bash tools/cli_timing.sh > $O/cli_timing.txt 2>&1; echo "cli rc=$?"
python3 tools/bench_server.py > $O/server_default.txt 2>&1; echo "server rc=$?"
GRAYHIP_PNG=fast python3 tools/bench_server.py > $O/server_fast.txt 2>&1; echo "server fast rc=$?"
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 2 > $O/reh2.json 2> $O/reh2.err; echo "reh2 rc=$?"
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 2 > $O/reh2_lb.json 2> $O/reh2_lb.err; echo "reh2_lb rc=$?"
python3 tools/bench_host_abi.py > $O/host_abi.txt 2>&1; echo "host abi rc=$?"
grep -v amdgpu.ids $O/cli_timing.txt | grep -v "^$" | head -80; cat $O/server_default.txt $O/server_fast.txt $O/host_abi.txt | grep -v amdgpu.ids
