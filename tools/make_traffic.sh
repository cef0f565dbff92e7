#!/bin/bash
# Regenerates profiles/traffic.json (HBM bytes per launch of the L4 hot-path kernels) and the counter summary
# profiles/<tag>_pmc_cost_volume_L4.txt on a GPU box:   tools/make_traffic.sh r02
# Counters are collected in passes of their own (tools/pmc.sh: --pmc + --kernel-trace only).
set -eu
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
tools/pmc.sh cv84 -- python3 tools/cv84_launch.py
python3 tools/traffic_from_pmc.py gpurun_out/pmc_cv84 profiles/traffic.json "$tag" \
    cost_volume_L4_bytes_per_launch="cost_volume_mfma_lds_kernel<false>" \
    warp_clamp_L4_bytes_per_launch=warp_nhwc_vec4_kernel \
    warp_cost_volume_L4_bytes_per_launch="cost_volume_mfma_lds_kernel<true>"
python3 tools/pmc_summary.py gpurun_out/pmc_cv84 "cost_volume_mfma_lds_kernel<false>" > profiles/${tag}_pmc_cost_volume_L4.txt
python3 tools/pmc_summary.py gpurun_out/pmc_cv84 "cost_volume_mfma_lds_kernel<true>" > profiles/${tag}_pmc_warp_cost_volume_L4.txt
python3 tools/pmc_summary.py gpurun_out/pmc_cv84 warp_nhwc_vec4_kernel > profiles/${tag}_pmc_warp_L4.txt
cp profiles/traffic.json gpurun_out/traffic.json
cp profiles/${tag}_pmc_*.txt gpurun_out/
