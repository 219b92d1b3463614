#!/bin/bash
# Wall-time anatomy of the drop-in CLI on the GPU box (developer tool): whole process vs decode / HIP start-up / encode.
set -e
IMG=${1:-tests/golden/test.jpg}
OUT=/tmp/cli_out.png
TIMEFORMAT="  wall %R s (user %U sys %S)"
for i in 1 2 3; do echo "bin/grayscale (reference-identical PNG bytes, the default):"; time bin/grayscale "$IMG" $OUT 1; done
md5sum $OUT
for i in 1 2 3; do echo "bin/grayscale GRAYHIP_PNG=fast:"; time GRAYHIP_PNG=fast bin/grayscale "$IMG" $OUT 1; done
export GRAYHIP_PNG=fast    # the level / thread sweep below concerns the band-parallel encoder
for lv in 1 3 6; do for th in 1 32; do echo "bin/grayscale GH_PNG_LEVEL=$lv GH_PNG_THREADS=$th:"; time GH_PNG_LEVEL=$lv GH_PNG_THREADS=$th bin/grayscale "$IMG" $OUT 1; ls -l $OUT | cut -d" " -f5; done; done
unset GRAYHIP_PNG
for i in 1 2; do echo "bin/grayscale_sobel:"; time bin/grayscale_sobel "$IMG" $OUT 1; done
echo "decode only (gh_imgtool):"; time bin/gh_imgtool decode "$IMG" /tmp/cli_raw.bin
python3 - <<'PY'
import ctypes, time, os
t0 = time.perf_counter()
lib = ctypes.CDLL(os.path.join("openmp-parallel-computing_amd", "lib", "libgrayhip.so"))
t1 = time.perf_counter()
rc = lib.gh_init(0)
t2 = time.perf_counter()
print(f"dlopen libgrayhip {1e3*(t1-t0):.0f} ms, gh_init (hipInit + context + first kernel module load) {1e3*(t2-t1):.0f} ms, rc={rc}")
PY
echo "--- resident server ---"
bin/grayhipd /tmp/gh_timing.sock > /tmp/grayhipd.log 2>&1 &
SRV=$!
for i in $(seq 1 50); do [ -S /tmp/gh_timing.sock ] && break; sleep 0.1; done
for i in 1 2 3 4; do echo "bin/grayscale via grayhipd:"; time GRAYHIP_SERVER=/tmp/gh_timing.sock bin/grayscale "$IMG" $OUT 1; done
kill $SRV; wait $SRV 2>/dev/null || true
GRAYHIP_PNG=fast bin/grayhipd /tmp/gh_timing2.sock > /tmp/grayhipd2.log 2>&1 &
SRV=$!
for i in $(seq 1 50); do [ -S /tmp/gh_timing2.sock ] && break; sleep 0.1; done
for i in 1 2 3 4; do echo "bin/grayscale via grayhipd started with GRAYHIP_PNG=fast:"; time GRAYHIP_SERVER=/tmp/gh_timing2.sock bin/grayscale "$IMG" $OUT 1; done
kill $SRV; wait $SRV 2>/dev/null || true
