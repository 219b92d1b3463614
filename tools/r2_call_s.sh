#!/bin/bash
# round 2, GPU call S (second form: at most 4 counters of a block per pass -- larger sets are refused and the aborting
# profiler then hangs until its timeout): memory-pipeline counters of the two ring kernels on an aligned (16384) and an unaligned (16370) row pitch
set -o pipefail
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2s
mkdir -p $O
SETS=(
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_LFIFO_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_READ_sum TCC_STREAMING_REQ_sum"
)
fails=0
for w in 16384 16370; do
  i=0
  for set in "${SETS[@]}"; do
    timeout -k 5 75 rocprofv3 --pmc $set -d $O/w$w/set$i -o pmc --output-format csv -- python3 tools/unaligned_counters.py $w 16384 12 > $O/w${w}_set$i.log 2>&1 || { echo "set $i w $w failed"; grep -m1 "error code" $O/w${w}_set$i.log; fails=$((fails+1)); [ $fails -ge 3 ] && exit 1; }
    i=$((i+1))
  done
  echo "== w=$w"; python3 tools/pmc_medians.py $O/w$w | tee $O/medians_$w.txt
done
find $O -name "*.csv" -size +3M -delete
