#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dq2; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for v in cur dqold cur; do
  if [ $v = cur ]; then unset HALVA_HIP_LIB; else export HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $OUT/$v -o b --output-format csv -- python3 tools/bench_sdpa.py > $OUT/$v.log 2>&1
  python3 - $OUT/$v <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'dq2' not in k: continue
    acc[k[:40]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items(): print(sys.argv[1].split('/')[-1], {a:round(b/1e6,2) for a,b in v.items()})
PY
done
rm -rf $OUT
export VARIANTS="dqold cur dqold cur" SKIPTESTS=
bash tools/r04/g20.sh 2>&1 | grep -v "dkv3\|fwd3\|delta"
