#!/bin/bash
# One GPU call that regenerates the round's evidence under profiles/ (and a copy under gpurun_out/profiles_<tag>/,
# which is what travels back from the GPU box):   tools/profile_round.sh r03
#   1. tools/make_traffic.sh: PMC passes (counters in runs of their own) -> profiles/traffic.json + <tag>_pmc_*.txt
#   2. tools/kbench.py at the level shapes of BASELINE configs 2, 4 and 5 -> <tag>_kbench*.json
#   3. bench.py under rocprofv3 --kernel-trace --stats -> <tag>_bench_under_rocprofv3.json, kernel stats, per-grid
#      durations, steady-state stats, one forward's timeline
#   4. bench.py (the line the driver will see; roofline.traffic now matches the kernel sources) -> <tag>_bench.json
set -u
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
out=gpurun_out/profiles_$tag
mkdir -p "$out"
bash tools/make_traffic.sh "$tag" > "$out/make_traffic.log" 2>&1 || echo "make_traffic failed"
PMC_PASSES="A B C G J" tools/pmc.sh sep -- python3 tools/sepbench.py --levels 4 --iters 2 > "$out/pmc_sep.log" 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_sep sepconv3x3_fused_kernel > profiles/${tag}_pmc_sepconv_fused_L4.txt 2>> "$out/pmc_sep.log"
python3 tools/kbench.py --json profiles/${tag}_kbench.json > "$out/kbench_c2.log" 2>&1
python3 tools/kbench.py --batch 16 --res 1024x2048 --levels 1,2,3,4 --iters 10 --json profiles/${tag}_kbench_config4.json > "$out/kbench_c4.log" 2>&1
python3 tools/kbench.py --batch 32 --dtype f16 --json profiles/${tag}_kbench_config5.json > "$out/kbench_c5.log" 2>&1
prof=$root/gpurun_out/prof_$tag
rm -rf "$prof"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$prof" -- \
    python3 "$root/bench.py" --no-extra --no-cpu-baseline --no-inflight --detail "" \
    > "$root/profiles/${tag}_bench_under_rocprofv3.json" 2> "$root/$out/rocprof_bench.err" ) || echo "rocprofv3 bench failed"
trace=$(find "$prof" -name '*kernel_trace.csv' | head -1)
stats=$(find "$prof" -name '*kernel_stats.csv' | head -1)
if [ -n "$trace" ]; then
  [ -n "$stats" ] && cp "$stats" profiles/${tag}_bench_kernel_stats.csv
  python3 tools/trace_steady.py "$trace" 20 60 profiles/${tag}_bench_steady_kernel_stats.csv > "$out/trace_steady.log" 2>&1
  python3 tools/by_grid.py "$trace" profiles/${tag}_bench_qpwc_kernels_by_grid.csv > "$out/by_grid.log" 2>&1
  python3 tools/ktrace.py "$prof" "$out/timeline.csv" 4000 > /dev/null 2>&1
  python3 tools/fwd_timeline.py "$out/timeline.csv" profiles/${tag}_forward_timeline.txt > /dev/null 2>&1
fi
python3 bench.py > profiles/${tag}_bench.json 2> "$out/bench.err"
cp bench_detail.json profiles/${tag}_bench_detail.json 2>/dev/null
cp profiles/${tag}_* profiles/traffic.json "$out/" 2>/dev/null
tail -c 600 profiles/${tag}_bench.json
