set -o pipefail
mkdir -p gpurun_out/r4_ring
python3 bench.py --no-cpu-baseline > gpurun_out/r4_ring/bench_ring3.json 2> gpurun_out/r4_ring/bench_ring3.err; echo "ring3 rc $?"
python3 bench.py --no-cpu-baseline --edge-ring 1 > gpurun_out/r4_ring/bench_ring1.json 2> gpurun_out/r4_ring/bench_ring1.err; echo "ring1 rc $?"
python3 bench.py --no-cpu-baseline > gpurun_out/r4_ring/bench_ring3b.json 2>/dev/null; echo "ring3b rc $?"
python3 bench.py --no-cpu-baseline --edge-ring 1 > gpurun_out/r4_ring/bench_ring1b.json 2>/dev/null; echo "ring1b rc $?"
for ring in 1 3 6; do
  echo "== tools/ab_fused.py --op fused --out-ring $ring: nontemporal (ships) vs plain vs sc1 output stores" >> gpurun_out/r4_ring/ab_store_policy.txt
  timeout -k 10 200 python3 tools/ab_fused.py --op fused --out-ring $ring --rounds 7 "nt=GH_FUSED_DMA_ST:2" "plain=GH_FUSED_DMA_ST:0" "sc1=GH_FUSED_DMA_ST:1" >> gpurun_out/r4_ring/ab_store_policy.txt 2>&1; echo "ab ring $ring rc $?"
done
cat gpurun_out/r4_ring/ab_store_policy.txt
python3 -c "
import json
for n in ('ring3','ring1','ring3b','ring1b'):
    d=json.load(open('gpurun_out/r4_ring/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['repeats']['ms_per_step'], d['kernels']['pipeline_pass_k']['ms'], d['parity_check'][:60])
"
