#!/bin/bash
# round-3 co-residency probe with the WHOLE library rebuilt with hipcc defaults (packed fp32 allowed) = round 3's condition
mkdir -p gpurun_out/r4cr
export BABE_HIP_LIB=$PWD/tools/abl_out/all_pk/libbabe_hip.so
export PARTNERS="bf16p fwd 256ch,bf16p units 256ch,bf16p vjp 256ch,bf16p units 64ch"
(echo "== whole library packed-built, mixed-radix FFT"; timeout 300 python3 tools/erratum/coresidency_probe.py 2>&1 | tail -12
 echo "== whole library packed-built, BABE_FFT_MIXED=0 (round 3's dense FFT as the rfft victim)"; BABE_FFT_MIXED=0 timeout 300 python3 tools/erratum/coresidency_probe.py 2>&1 | tail -12) > gpurun_out/r4cr/probe_all_pk.txt
cat gpurun_out/r4cr/probe_all_pk.txt
