#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the round's committed measurements -- box read rate, shard sweep, SURVEY 8(d) input sets, the
# small-batch L2 kernel (part 1), and the rocprofv3 passes of the default bench + the SQ counters (part 2).  Two gpurun calls:
# one call is limited to 20 minutes.
# usage: tools/run_round_measurements.sh <round-tag> [1|2]
TAG=${1:-r05}
PART=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
if [ "$PART" = "1" ]; then
  ./tools/micro/readbw 5 40 > gpurun_out/$TAG/readbw.txt 2>&1
  timeout -k 10 400 python tools/shard_sweep.py 8 4 2 1 flags=0 flags=2048 > gpurun_out/$TAG/shard_sweep.json 2> gpurun_out/$TAG/shard_sweep.err; echo "sweep rc=$?"
  timeout -k 10 600 python tools/measure_all.py gpurun_out/$TAG/measurements.json > gpurun_out/$TAG/measure_all.log 2>&1; echo "measure rc=$?"
  timeout -k 10 150 python tools/l2diff_probe.py > gpurun_out/$TAG/l2diff.txt 2>&1; echo "l2diff rc=$?"
else
  bash tools/collect_profiles.sh $TAG; echo "profiles rc=$?"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq_$TAG -o s -- python3 bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/pmc_sq_$TAG.json 2> gpurun_out/pmc_sq_$TAG.err; echo "sq rc=$?"
  python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq_$TAG/*/s_counter_collection.csv 2.0 > gpurun_out/$TAG/pmc_sq_summary.txt 2>&1 || python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq_$TAG/s_counter_collection.csv 2.0 > gpurun_out/$TAG/pmc_sq_summary.txt 2>&1; echo "sq summary rc=$?"
  rm -rf gpurun_out/pmc_sq_$TAG
fi
