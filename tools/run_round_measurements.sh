#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the round's committed measurements in one call -- box read rate, shard sweep,
# SURVEY 8(d) input sets, the small-batch L2 kernel, and the rocprofv3 passes of the default bench.
# usage: tools/run_round_measurements.sh <round-tag>
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
./tools/micro/readbw 5 40 > gpurun_out/$TAG/readbw.txt 2>&1
timeout -k 10 500 python tools/shard_sweep.py 8 4 2 1 flags=0 flags=2048 > gpurun_out/$TAG/shard_sweep.json 2> gpurun_out/$TAG/shard_sweep.err; echo "sweep rc=$?"
timeout -k 10 900 python tools/measure_all.py gpurun_out/$TAG/measurements.json > gpurun_out/$TAG/measure_all.log 2>&1; echo "measure rc=$?"
timeout -k 10 200 python tools/l2diff_probe.py > gpurun_out/$TAG/l2diff.txt 2>&1; echo "l2diff rc=$?"
bash tools/collect_profiles.sh $TAG; echo "profiles rc=$?"
