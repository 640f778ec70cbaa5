#!/bin/bash
# Run ON THE GPU BOX: only the BASELINE config 5 parts of tools/collect_profiles.sh (steps 1b and 3b), for a round whose
# changes touch the 16-bit kernels alone.   tools/collect_profiles_config5.sh r05 <commit>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
COMMIT=${2:-unknown}
OUT=gpurun_out/profiles
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_c5 -- python3 bench.py --dtype f16 --width 2048 --height 1024 --steps 100 --warmup 3 --no-cpu-baseline --no-cold --probe-steps 2 > $OUT/${TAG}_bench_under_rocprof_config5_f16.json 2> $OUT/_c5.err
cp "$(find $OUT/_c5 -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_config5_f16.csv
python3 tools/trace_by_grid.py "$(find $OUT/_c5 -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_kernel_trace_by_grid_config5_f16.csv
rm -rf $OUT/_pmc_*
for L in 0 1 2 3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_pmc_L${L}_$name -- python3 tools/prof_kernels.py --levels $L --warp --reps 5 --dtype f16 --width 2048 --height 1024 > /dev/null 2> $OUT/_pmc_L${L}_$name.err
  done
done
python3 tools/pmc_to_traffic.py $OUT ${TAG}_config5_f16 $COMMIT
rm -f $OUT/${TAG}_config5_f16_pmc_counters.csv
rm -rf $OUT/_c5 $OUT/_pmc_*
ls -la $OUT | head -20
