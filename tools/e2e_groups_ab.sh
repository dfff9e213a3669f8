export KNN355_LIB=$PWD/knn-for-homology_amd/libknn355_trace.so
for cfg in "256 8" "32 4" "32 5" "32 6" "32 8" "256 8" "32 4" "32 6"; do
  set -- $cfg
  echo "== MIN_MB=$1 GROUPS=$2"
  KNN355_SELF_STREAM_MIN_MB=$1 KNN355_SELF_GROUPS=$2 timeout -k 10 120 python tools/e2e_probe.py 2>&1 | grep -E "end to end|search_self" || exit 1
done
