#!/bin/bash
# Regenerates profiles/traffic.json (HBM bytes per launch of the L4 hot-path kernels at the shapes of BASELINE configs
# 2, 4 and 5) and the counter summaries profiles/<tag>_pmc_*.txt on a GPU box:   tools/make_traffic.sh r03
# Counters are collected in passes of their own (tools/pmc.sh: --pmc + --kernel-trace only).  Config 2 gets every
# pass (SQ / LDS / TCP / TCC counters for the summaries), configs 4 and 5 the cache and HBM-traffic passes.
set -eu
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
rm -f profiles/traffic.json
for cfg in 2 4 5; do
  if [ "$cfg" = 2 ]; then passes=""; pre=""; else passes="C D E F H I"; pre="c${cfg}_"; fi
  PMC_PASSES="$passes" tools/pmc.sh cv84_c$cfg -- python3 tools/cv84_launch.py --config $cfg
  if [ "$cfg" = 5 ]; then cv=cost_volume_mfma_lds_f16_kernel; else cv=cost_volume_mfma_lds_kernel; fi
  python3 tools/traffic_from_pmc.py gpurun_out/pmc_cv84_c$cfg profiles/traffic.json "$tag" \
      ${pre}cost_volume_L4_bytes_per_launch="${cv}<false>" \
      ${pre}warp_clamp_L4_bytes_per_launch=warp_nhwc \
      ${pre}warp_cost_volume_L4_bytes_per_launch="${cv}<true>|lds8x16_warp_kernel|lds16_kernel<true>" || true
  sfx=""; [ "$cfg" = 2 ] || sfx="_c$cfg"
  python3 tools/pmc_summary.py gpurun_out/pmc_cv84_c$cfg "kernel<false>" > profiles/${tag}_pmc_cost_volume_L4$sfx.txt
  python3 tools/pmc_summary.py gpurun_out/pmc_cv84_c$cfg "kernel<true>|lds8x16_warp_kernel" > profiles/${tag}_pmc_warp_cost_volume_L4$sfx.txt
  python3 tools/pmc_summary.py gpurun_out/pmc_cv84_c$cfg warp_nhwc > profiles/${tag}_pmc_warp_L4$sfx.txt
done
cp profiles/traffic.json gpurun_out/traffic.json
cp profiles/${tag}_pmc_*.txt gpurun_out/
