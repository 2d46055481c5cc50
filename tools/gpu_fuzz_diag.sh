L=gpurun_out/r05_fuzz_diag.log; : > $L
d() { first=$1; t=$2; shift 2; echo "== trial $t $*" >> $L; env LG_FUZZ_FIRST=$first LG_FUZZ_N2=0 "$@" python tools/gpu_fuzz.py $((t - first + 1)) $t >> $L 2>&1; }
for t in 3589 3635 3709 3802 3931; do d 150 $t; done
for t in 4987 5397 6430; do d 4650 $t LG_FUZZ_SEG=64 LG_FUZZ_LONG=parallel; done
d 6900 7626 LG_FUZZ_SEG=64 LG_FUZZ_LONG=serial
for t in 8748 9086 9192 9243 9431 9566; do d 8400 $t LG_FUZZ_SYNC=off; done
grep -v "^fuzz:" $L | cut -c1-400
