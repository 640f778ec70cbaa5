#!/bin/bash
# GPU box: SQ / traffic counters of the warp kernels on one level (separate --pmc passes).
# usage: tools/pmc_warp16.sh <outfile> <dtype> <5|3> <level>      (PROF=tools/prof_corr16.py MATCH=corr: the correlation kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUTF=$1; shift
OUT=gpurun_out/_pmcw16
rm -rf $OUT; mkdir -p $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_p_$name -- python3 ${PROF:-tools/prof_warp16.py} "$@" > /dev/null 2> $OUT/_p_$name.err || echo "pass $name failed" >&2
done
python3 - $OUT ${MATCH:-warp} ${PROF:-tools/prof_warp16.py} "$@" > $OUTF <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
print("# rocprofv3 --pmc over %s (per launch, mean of n)" % " ".join(sys.argv[3:]))
match = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/_p_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if match not in k: continue
        acc[k.replace("void cerb::(anonymous namespace)::", "").split("(")[0][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUTF
rm -rf $OUT
