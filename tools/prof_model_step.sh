#!/bin/bash
# GPU box: rocprofv3 kernel stats of the host model's training step (bench.py --step model): where its time goes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
mkdir -p gpurun_out/profiles
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profiles/_model -- python3 bench.py --step model --steps 6 --warmup 2 > gpurun_out/profiles/${TAG}_bench_under_rocprof_step_model.json 2> gpurun_out/profiles/_model.err
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
p = glob.glob("gpurun_out/profiles/_model/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(p))]
rows.sort()
# one step = two launches of the 256-wide strip backward (one per flow direction): the window of the LAST FOUR steps
# starts at the first kernel after the 9th-from-last such launch (MIOpen's find-mode kernels of the first steps stay out)
marks = [i for i, r in enumerate(rows) if "corr_bwd_d4_strip_kernel" in r[2] and "StripCfg<64" in r[2]]
lo = marks[-9] + 1
win = rows[lo:]
steps = 4.0
acc = collections.defaultdict(lambda: [0, 0])
for _, d, n in win:
    key = n.replace("void cerb::(anonymous namespace)::", "")[:100]
    acc[key][0] += 1; acc[key][1] += d
tot = sum(v[1] for v in acc.values())
span = (win[-1][0] + win[-1][1] - win[0][0])
ours = {k: v for k, v in acc.items() if "at::native" not in k and any(t in k for t in ("corr_fwd", "corr_bwd", "corr_grad_prep", "warp_fwd", "warp_bwd", "warp_fewc", "upsample_fwd_kernel", "upsample_bwd_kernel", "area_resize", "area_pyramid", "warp_context"))}
top = sorted(acc.items(), key=lambda kv: -kv[1][1])
out = open("gpurun_out/profiles/%s_kernel_stats_step_model.csv" % tag, "w")
out.write("# rocprofv3 --kernel-trace of `bench.py --step model` (HRNetV2-W32 + PWC flow head training step, 4 pairs, fp32): the LAST FOUR steps of the run\n")
out.write("# (MIOpen's find-mode kernels of the first steps excluded); per step: launches, ms of kernel time, share\n")
out.write("kernel,launches_per_step,ms_per_step,percent\n")
for k, v in top[:20]:
    out.write('"%s",%.1f,%.3f,%.2f\n' % (k, v[0] / steps, v[1] / steps / 1e6, 100.0 * v[1] / tot))
out.write("# --- every kernel of this package ---\n")
for k, v in sorted(ours.items(), key=lambda kv: -kv[1][1]):
    out.write('"%s",%.1f,%.4f,%.3f\n' % (k, v[0] / steps, v[1] / steps / 1e6, 100.0 * v[1] / tot))
so = sum(v[1] for v in ours.values())
out.write("# kernel time per step %.1f ms in %.0f launches; wall per step in this window %.1f ms; this package's kernels %.3f ms per step = %.2f %% of the kernel time\n"
          % (tot / steps / 1e6, sum(v[0] for v in acc.values()) / steps, span / steps / 1e6, so / steps / 1e6, 100.0 * so / tot))
out.close()
print(open("gpurun_out/profiles/%s_kernel_stats_step_model.csv" % tag).read())
PY
rm -rf gpurun_out/profiles/_model gpurun_out/profiles/_model.err
