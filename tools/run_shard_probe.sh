mkdir -p gpurun_out/r3f && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./tools/micro/readbw 5 40 > gpurun_out/r3f/readbw.txt 2>&1
timeout -k 10 300 python -m pytest tests/test_flat_gpu.py -x -q -m gpu -k "tile_minimum or tiling or edge or tie or sorted" > gpurun_out/r3f/flat.log 2>&1; tail -2 gpurun_out/r3f/flat.log
timeout -k 10 300 python tools/shard_sweep.py 8 1 flags=0 flags=2 flags=2048 > gpurun_out/r3f/sweep.json 2> gpurun_out/r3f/sweep.err
KNN355_LIB=$GRAFT_REPO_ROOT/knn-for-homology_amd/libknn355_trace.so timeout -k 10 200 python tools/wg_timeline.py 1250000 0 > gpurun_out/r3f/tl_pub.txt 2>&1
cat gpurun_out/r3f/readbw.txt; cut -c1-250 gpurun_out/r3f/sweep.err | tail -6; cat gpurun_out/r3f/tl_pub.txt
