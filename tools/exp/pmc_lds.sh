#!/bin/bash
# LDS bank conflicts of the mixed-radix passes: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel at n = $1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
n=${1:-320}
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmclds$n -o p --output-format csv -- python3 tools/pmc_pass.py $n 1 > gpurun_out/pmclds$n.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmclds$n/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r['Kernel_Name'][-60:]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'pass' not in k: continue
    print('lane-instr per point', round(v.get('SQ_INSTS_VALU',0)*64/($n*$n),1), k, {c: int(x) for c, x in v.items()}, 'conflict/active', round(v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1), 3))
PY
