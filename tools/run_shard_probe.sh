mkdir -p gpurun_out/r3f && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./tools/micro/readbw > gpurun_out/r3f/readbw.txt 2>&1
timeout -k 10 300 python tools/shard_sweep.py 8 flags=0 flags=2048 > gpurun_out/r3f/sweep.json 2> gpurun_out/r3f/sweep.err
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3f/prof1 -o s1 -- python3 bench.py --nb-total 1250000 --no-cpu --no-batch --no-extras --steps 60 --warmup 40 > gpurun_out/r3f/b1.json 2> gpurun_out/r3f/b1.err
python tools/shard_timeline.py gpurun_out/r3f/prof1 2 > gpurun_out/r3f/tl1.txt 2>&1
cat gpurun_out/r3f/readbw.txt; cut -c1-250 gpurun_out/r3f/sweep.err | tail -3; cat gpurun_out/r3f/tl1.txt
