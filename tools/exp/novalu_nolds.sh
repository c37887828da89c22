V=$PWD/fdes_amd/csrc/build/variants
for v in tree novalu nolds noboth; do
  if [ $v = tree ]; then L=$PWD/fdes_amd/csrc/libFDES_SHARED_LIB.so; else L=$V/lib_$v.so; fi
  FDES_LIB=$L python tools/bench_one.py 2048 2 5 1 6 2>/dev/null
  FDES_LIB=$L python tools/bench_one.py 2048 1 6 2 1 2>/dev/null
  FDES_LIB=$L python tools/bench_one.py 4096 2 5 1 6 2>/dev/null
done
tools/ab_libs.sh "tree novalu nolds noboth" 1
