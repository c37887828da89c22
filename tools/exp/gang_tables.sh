#!/bin/bash
# The tables DESIGN 4.2 quotes for gangs (one box, one run): SrTiO3 series with 16 frozen-phonon configurations per tilt at
# 256^2 .. 1024^2 without gangs (three lanes), with the automatic choice and around it; series with one configuration per
# measurement through the boundary call; the five BASELINE configurations.
for n in 128 256 400 512; do
  N=$n FRPH=16 GANG=0 python tools/bench_c4.py 3
  N=$n FRPH=16 python tools/bench_c4.py 0
  for g in 4 8 16; do N=$n FRPH=16 GANG=$g python tools/bench_c4.py 1 2; done
done
python tools/bench_series.py
python tools/exp/series_big.py 512 128
python tools/exp/series_big.py 256 256
python tools/bench_configs.py
