#!/bin/bash
# Fuzz trials beyond the round script's 0..149 (tools/gpu_fuzz.py, LG_FUZZ_FIRST), under the variant switches the library has:
# segmented backward with 64-entry segments, long-tile walks serial / parallel (colour forward AND the significance pass's
# parallel walk), the host-synchronous forward, 40-bit keys.  Usage: gpu_fuzz_campaign.sh [K] [FIRST].  Log: gpurun_out/r06_fuzz_campaign[_from_FIRST].log
mkdir -p gpurun_out
L=gpurun_out/r06_fuzz_campaign${2:+_from_$2}.log
: > $L
run() { echo "== $*" >> $L; ( time env "$@" ) >> $L 2>&1; }
K=${1:-1}        # scale: K = 1 is ~1 minute of GPU time (a trial takes ~70 ms), K = 15 ~12 minutes
B=${2:-150}      # first trial number (the round script covers 0..149; the round-5 campaign ran K = 15 from 150)
T="timeout -s KILL 900 python tools/gpu_fuzz.py"
run LG_FUZZ_FIRST=$B LG_FUZZ_N3=$((150 * K)) $T $((300 * K))
run LG_FUZZ_FIRST=$((B + 300 * K)) LG_FUZZ_SEG=64 LG_FUZZ_LONG=parallel $T $((150 * K))
run LG_FUZZ_FIRST=$((B + 450 * K)) LG_FUZZ_SEG=64 LG_FUZZ_LONG=serial $T $((100 * K))
run LG_FUZZ_FIRST=$((B + 550 * K)) LG_FUZZ_SYNC=off $T $((100 * K))
run LG_FUZZ_FIRST=$((B + 650 * K)) LG_FUZZ_NARROW=1 LG_FUZZ_N2=$((100 * K)) $T 20
grep -E "^==|fuzz:|MISMATCH|real|Error|error" $L | cut -c1-260
