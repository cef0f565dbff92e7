#!/bin/bash
# Evidence for the opt-in bf16x3 arithmetic, one GPU call:   tools/profile_x3.sh r03
#   1. bench.py --matmul bf16x3 under rocprofv3 --kernel-trace --stats -> <tag>_x3_bench_under_rocprofv3.json, steady-state
#      kernel stats, one forward's timeline
#   2. PMC passes A / B / G over tools/x3bench.py and tools/sepx3bench.py -> <tag>_pmc_x3.txt
#   3. tools/x3bench.py, tools/sepx3bench.py, tools/x3_epe.py -> <tag>_x3_kernels.txt (times and errors vs float64 / oracle)
set -u
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
out=gpurun_out/profiles_x3_$tag
mkdir -p "$out"
prof=$root/gpurun_out/prof_x3_$tag
rm -rf "$prof"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$prof" -- \
    python3 "$root/bench.py" --matmul bf16x3 --no-extra --no-cpu-baseline --no-inflight --detail "" \
    > "$root/$out/${tag}_x3_bench_under_rocprofv3.json" 2> "$root/$out/rocprof_bench.err" ) || echo "rocprofv3 bench failed"
trace=$(find "$prof" -name '*kernel_trace.csv' | head -1)
if [ -n "$trace" ]; then
  python3 tools/trace_steady.py "$trace" 20 60 "$out/${tag}_x3_bench_steady_kernel_stats.csv" > "$out/trace_steady.log" 2>&1
  python3 tools/ktrace.py "$prof" "$out/timeline.csv" 4000 > /dev/null 2>&1
  python3 tools/fwd_timeline.py "$out/timeline.csv" "$out/${tag}_x3_forward_timeline.txt" > /dev/null 2>&1
fi
PMC_PASSES="A B G" bash tools/pmc.sh x3enc -- python3 tools/x3bench.py > "$out/pmc_x3enc.log" 2>&1
PMC_PASSES="A B G" bash tools/pmc.sh x3sep -- python3 tools/sepx3bench.py --iters 4 > "$out/pmc_x3sep.log" 2>&1
{ echo "# rocprofv3 PMC (tools/pmc.sh, passes A / B / G) over tools/x3bench.py and tools/sepx3bench.py: the bf16x3 kernels and"
  echo "# their fp32-instruction twins, mean per dispatch (a kernel's launches at several shapes are averaged together)"
  python3 tools/pmc_summary.py gpurun_out/pmc_x3enc "conv3x3_mish"
  python3 tools/pmc_summary.py gpurun_out/pmc_x3sep "sepconv3x3"; } > "$out/${tag}_pmc_x3.txt" 2>&1
{ python3 tools/x3bench.py; python3 tools/sepx3bench.py; python3 tools/x3_epe.py; } 2>&1 | grep -v amdgpu.ids > "$out/${tag}_x3_kernels.txt"
rm -f "$out/timeline.csv"
tail -c 300 "$out/${tag}_x3_bench_under_rocprofv3.json"
