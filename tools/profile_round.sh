#!/bin/bash
# Collects the profiles of a round on the GPU box into gpurun_out/prof_rNN/ (copy what is to be judged into profiles/):
#   kernel statistics (rocprofv3 --kernel-trace --stats) of the headline bench (2048^2), of C5 (4096^2) and of 1024^2,
#   PMC traffic and SQ counters (one run per counter group, never combined with tracing domains other than the kernel
#   trace) for 2048^2 and 4096^2.            usage: tools/profile_round.sh r02
set -e
R=${1:-r05}
O=$PWD/gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
Q=$GRAFT_REPO_ROOT
[ -z "$Q" ] && Q=/root/repo
B2048="python3 $Q/bench.py --steps 4 --warmup 1 --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0"
B1024="python3 $Q/bench.py --size 1024 --slices 64 --steps 8 --warmup 2 --cpu-baseline 0 --extra-skip-run 0 --extras 0"
C5="python3 $Q/tools/run_c5.py 2"
echo "== kernel traces"
rocprofv3 --kernel-trace --stats -d $O/kt2048 -o kt -- $B2048 > $O/kt2048.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt4096 -o kt -- $C5 > $O/kt4096.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt1024 -o kt -- $B1024 > $O/kt1024.log 2>&1
B1000="python3 $Q/bench.py --size 1000 --slices 64 --steps 8 --warmup 2 --cpu-baseline 0 --extra-skip-run 0 --extras 0"
rocprofv3 --kernel-trace --stats -d $O/kt1000 -o kt -- $B1000 > $O/kt1000.log 2>&1
for s in 2048 4096 1024 1000; do python3 $Q/tools/rocpd_stats.py $(ls $O/kt$s/*.db $O/kt$s/*/*.db 2>/dev/null | head -1) > $O/${R}_kernel_stats_$s.csv; done
echo "== PMC"
P2048="python3 $Q/bench.py --steps 1 --warmup 0 --lanes 1 --slices 32 --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0"
P4096="python3 $Q/tools/run_c5.py 1 lanes=1 slices=16"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc2048/g$i -o p -- $P2048 > $O/pmc2048_g$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc4096/g$i -o p -- $P4096 > $O/pmc4096_g$i.log 2>&1
  echo "   group $i done"
done
# memory-side view (round 3): L2 hit / miss and the fabric (EA) requests behind it; which of these exist on this box is
# recorded in $O/counters_available.txt.  A group that the profiler refuses is skipped, not fatal.
rocprofv3 -L > $O/counters_all.txt 2>&1 || true
grep -i -E "TCC_HIT|TCC_MISS|TCC_EA0?_(RD|WR)REQ|MALL|TCC_REQ|TCC_BUBBLE|HBM" $O/counters_all.txt | head -80 > $O/counters_available.txt || true
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc2048/g$i -o p -- $P2048 > $O/pmc2048_g$i.log 2>&1 || echo "   group $i ($grp) refused at 2048"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc4096/g$i -o p -- $P4096 > $O/pmc4096_g$i.log 2>&1 || echo "   group $i ($grp) refused at 4096"
  echo "   group $i done"
done
cd $Q
python3 tools/pmc_summary.py $O/${R}_pmc.json $O/${R}_pmc_sq.md "2048x2048 (bench.py, 1 lane, 32 slices)=$O/pmc2048" "4096x4096 (C5 specimen, 1 lane, 16 slices)=$O/pmc4096"
ls -la $O
