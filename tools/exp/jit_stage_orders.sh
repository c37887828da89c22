#!/bin/bash
# usage: tools/exp/jit_stage_orders.sh SIZE "a,b,c[,d]" ...   slice-propagations/s of the C3 specimen at SIZE^2 with the row passes compiled for each
# stage order in turn (FDES_JIT_STAGES, gen_jit.cpp); "auto" = the automatic choice
s=$1; shift
for o in "$@"; do
  if [ "$o" = auto ]; then unset FDES_JIT_STAGES; else export FDES_JIT_STAGES=$o; fi
  echo -n "$o : "; tools/exp/bench_sizes.sh "$s"
done
