#!/bin/bash
# A/B of tuning flags (KNN355_FLAGS) on the bench's batch legs and the streaming step: tools/ab_flags.sh 0 2048 0 2048 (needs the developer build: make -C knn-for-homology_amd/csrc trace, KNN355_LIB=.../libknn355_trace.so -- the shipped library ignores KNN355_FLAGS)
for fl in "$@"; do
  echo "== KNN355_FLAGS=$fl"
  KNN355_FLAGS=$fl timeout -k 10 300 python bench.py --no-extras --no-cpu --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = o['batch']; a = o['all_vs_all_query_sharded']
print('stream ms/step', round(o['ms_per_step'], 4), 'scan', round(o['roofline']['avg_kernel_ms'], 4))
print('cath kernel', round(b['kernel_ms'], 4), 'search', round(b['ms'], 4), 'frac', round(b['roofline']['frac'], 4), '| self kernel', round(b['self_search']['kernel_ms'], 4), 'search', round(b['self_search']['ms'], 4), '| e2e', round(b['end_to_end']['ms'], 3))
print('pfam all-vs-all ms', round(a['ms'], 2), 'frac', round(a['roofline']['frac'], 4), '| symmetric', round(a['self_search_symmetric']['ms'], 2))
" || exit 1
done
