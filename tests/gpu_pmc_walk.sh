#!/bin/bash
# PMC digests of the gravity walks of tests/gpu_walk_only.py for one library variant:
#   tests/gpu_pmc_walk.sh <variant.so | ""> <tag>
R=$PWD
SO=$1; TAG=$2
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
[ -n "$SO" ] && export GHIP_LIBGHIP=$R/$SO
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $R/tests/gpu_walk_only.py 64 2 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_grav_walk<0" in n: k="newton"
        elif "k_grav_walk<2" in n: k="ewald"
        else: continue
        a=acc[(k,r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for (k,c),a in sorted(acc.items()):
    print("$TAG %-7s %-26s %.4e (%d launches incl. the BH pass)"%(k,c,a[0]/a[1],a[1]))
PY
