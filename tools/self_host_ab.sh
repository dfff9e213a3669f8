for sz in "14433 301 9" "30000 301 7" "60000 100 5" "200000 100 3" "200000 1000 3"; do
  for st in 0 1 0 1; do
    echo -n "stream=$st  "
    KNN355_SELF_STREAM=$st timeout -k 10 200 python tools/self_host_probe.py $sz || exit 1
  done
done
