#!/bin/bash
# GPU box: rocprofv3 durations + fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) of the loss-side kernels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=gpurun_out/profiles
mkdir -p $OUT
FAILED=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_ls -- python3 tools/prof_loss_side.py 8 > /dev/null 2> $OUT/_ls.err || { echo "rocprofv3 trace pass failed: $OUT/_ls.err" >&2; FAILED=1; }
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_lspmc_$name -- python3 tools/prof_loss_side.py 4 > /dev/null 2> $OUT/_lspmc_$name.err || { echo "rocprofv3 --pmc $pass failed: $OUT/_lspmc_$name.err" >&2; FAILED=1; }
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
want = ("area_pyramid_kernel", "warp_fewc_kernel", "corr_grad_prep_kernel", "corr_bwd_d4_strip_kernel")
dur = collections.defaultdict(list)
for p in glob.glob(out + "/_ls/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        for w in want:
            if w in n:
                dur[(w, n)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/_lspmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        for w in want:
            if w in n:
                cnt[(w, n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
alg = {"area_pyramid_kernel": 4 * 12 * (512 * 1024 + 256 * 512 + 128 * 256 + 64 * 128), "corr_grad_prep_kernel": 3 * 4 * 81 * 128 * 256 * 4, "corr_bwd_d4_strip_kernel": 109576192}
f = open("%s/%s_loss_side_kernels.csv" % (out, tag), "w")
f.write("# rocprofv3 over tools/prof_loss_side.py (4 pairs, 512 x 1024 frames, every launch on fresh tensors): kernel durations (us, mean of n),\n")
f.write("# fabric traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (separate --pmc passes), SQ counters per launch\n")
f.write("kernel,n,avg_us,min_us,traffic_MB,algorithmic_MB,frac_of_8TBps,SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,TCC_HIT_sum,TCC_MISS_sum\n")
for (w, n), d in sorted(dur.items()):
    c = cnt.get((w, n), {})
    m = lambda k: (sum(c[k]) / len(c[k])) if k in c and c[k] else float("nan")
    traffic = (2 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024 / 1e6
    isb = "Lb1E" in n or ", true>" in n
    a = alg.get(w)
    if w == "warp_fewc_kernel":
        a = 83886080 if isb else 67108864
    avg = sum(d) / len(d) / 1e3
    short = n.split("(")[0].replace("void cerb::(anonymous namespace)::", "")[:90]
    f.write('"%s",%d,%.2f,%.2f,%.1f,%.1f,%.3f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f\n' % (short, len(d), avg, min(d) / 1e3, traffic, a / 1e6, a / (avg * 1e-6) / 8e12,
            m("SQ_WAVES"), m("SQ_INSTS_VALU"), m("SQ_INSTS_VMEM_RD"), m("SQ_INSTS_VMEM_WR"), m("TCC_HIT_sum"), m("TCC_MISS_sum")))
f.close()
print(open("%s/%s_loss_side_kernels.csv" % (out, tag)).read())
PY
# a failed pass keeps its log (and the exit code says so); only this script's own scratch is removed
if [ $FAILED -ne 0 ]; then exit 1; fi
rm -rf $OUT/_ls $OUT/_lspmc_* $OUT/_ls.err $OUT/_lspmc_*.err
