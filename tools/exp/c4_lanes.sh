for q in unset 8; do
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "GPU_MAX_HW_QUEUES=$q"
  FRPH=16 timeout -k 10 300 python tools/bench_c4.py 3 4 6 8 || exit 1
done
