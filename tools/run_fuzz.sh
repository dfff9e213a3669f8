#!/bin/bash
# Runs ON THE GPU BOX: the three developer fuzzers on the final kernels, time-bounded.  usage: tools/run_fuzz.sh <out file> [seed base]
OUT=${1:-gpurun_out/fuzz.txt}
S=${2:-51}
{
  echo "# tests/fuzz_gpu.py (two seeds, 220 s of cases each), tests/fuzz_stream_gpu.py, tests/fuzz_sym_gpu.py on the final kernels; flags now include the 256 x 256 tile (524288, query_tile 256)"
  timeout -k 10 300 python tests/fuzz_gpu.py 5000 $S 220 | grep -E "FAIL|FUZZ" ; echo "fuzz_gpu seed $S (220 s of cases)"
  timeout -k 10 300 python tests/fuzz_gpu.py 5000 $((S+1)) 220 | grep -E "FAIL|FUZZ" ; echo "fuzz_gpu seed $((S+1)) (220 s of cases)"
  timeout -k 10 200 python tests/fuzz_stream_gpu.py 150 $((S+2)) | grep -E "FAIL|FUZZ" ; echo "fuzz_stream_gpu seed $((S+2)): rc=$?"
  timeout -k 10 240 python tests/fuzz_sym_gpu.py 120 $((S+3)) | grep -E "FAIL|FUZZ" ; echo "fuzz_sym_gpu seed $((S+3)): rc=$?"
} > $OUT 2>&1
