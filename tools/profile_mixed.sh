#!/bin/bash
# Round 5: kernel statistics and HBM traffic counters of the fused loop on the grid sizes a .qsc gives beyond 2048 (mixed-radix
# rows, fft_gen.hip): 3000^2 and 4000^2 (m = 2 nx, src/rwQsc.cu:943-948).  Output: gpurun_out/prof_$R/ (copy into profiles/).
#   usage: tools/profile_mixed.sh r05 "3000 4000"
set -e
R=${1:-r05}
SIZES=${2:-"3000 4000"}
O=$PWD/gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
Q=$GRAFT_REPO_ROOT
[ -z "$Q" ] && Q=/root/repo
for s in $SIZES; do
  B="python3 $Q/bench.py --size $s --slices 32 --steps 4 --warmup 1 --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0"
  rocprofv3 --kernel-trace --stats -d $O/kt$s -o kt -- $B > $O/kt$s.log 2>&1
  python3 $Q/tools/rocpd_stats.py $(ls $O/kt$s/*.db $O/kt$s/*/*.db 2>/dev/null | head -1) > $O/${R}_kernel_stats_$s.csv
  P="python3 $Q/bench.py --size $s --slices 16 --steps 1 --warmup 0 --lanes 1 --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0"
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$s/g$i -o p -- $P > $O/pmc${s}_g$i.log 2>&1
  done
  echo "   $s done"
done
cd $Q
args=""
for s in $SIZES; do args="$args ${s}x${s}(bench.py,1lane,16slices)=$O/pmc$s"; done
python3 tools/pmc_summary.py $O/${R}_pmc_mixed.json $O/${R}_pmc_sq_mixed.md $args
ls $O
