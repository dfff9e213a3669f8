#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats of the default bench
# command, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots": the two do not fit one pass).  Outputs land in gpurun_out/.
# usage: tools/collect_profiles.sh <round-tag>
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py > gpurun_out/bench_prof_$TAG.json 2> gpurun_out/bench_prof_$TAG.err
echo "stats pass ok"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_fetch_$TAG.json 2> gpurun_out/pmc_fetch_$TAG.err
echo "fetch pass ok"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_write_$TAG.json 2> gpurun_out/pmc_write_$TAG.err
echo "write pass ok"
