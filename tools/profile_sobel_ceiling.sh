#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the counters behind "what bounds the plane Sobel" -- tools/sobel_ceiling.py under rocprofv3, one
# counter group per pass (never combined with a trace), every variant's kernel symbol in each pass.  Every pass under its own timeout
# (a counter set the hardware cannot collect makes rocprofv3 abort and then sit in its signal handler) and reported as it ends.
#   usage: tools/profile_sobel_ceiling.sh <out dir under gpurun_out/> [passes...]      (default: all)
set -o pipefail
OUT=${1:-gpurun_out/r4_sobel_ceiling}
shift || true
WANT=${*:-timing stats ea_rd ea_wr tcc tcp ta1 ta2 tlb sq lds}
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $WANT " == *" $1 "* ]]; }
if has timing; then timeout -k 10 200 python3 ${GH_CEILING_TOOL:-tools/sobel_ceiling.py} --rounds 3 > "$OUT/timing.txt" 2> "$OUT/timing.err"; echo "timing rc $?"; fi
if has stats; then
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 ${GH_CEILING_TOOL:-tools/sobel_ceiling.py} --launches 100 > "$OUT/stats.txt" 2> "$OUT/stats.err"; echo "stats rc $?"
fi
pass() {   # name, counters...
  local name=$1; shift
  has "$name" || return 0
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 ${GH_CEILING_TOOL:-tools/sobel_ceiling.py} --launches 40 > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err"
  echo "pmc $name rc $?"
}
pass ea_rd   TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL
pass ea_wr   TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL
pass tcc     TCC_BUSY TCC_REQ TCC_HIT TCC_MISS
pass tcp     TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ
pass ta1     TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES
pass ta2     TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS
pass tlb     TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TOTAL_CACHE_ACCESSES
pass sq      SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass lds     SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 tools/pmc_medians.py "$OUT" > "$OUT/medians.txt"
echo "sobel ceiling profile complete"
