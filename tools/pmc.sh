#!/bin/bash
# PMC counter passes with rocprofv3 (counters in their own runs, no trace domains
# besides --kernel-trace).  Usage: tools/pmc.sh <tag> -- <program> [args...]
# Results: gpurun_out/pmc_<tag>/<pass>/.../*_counter_collection.csv
# The program's arguments are resolved against the repo root BEFORE the cd to /tmp (rocprofv3 wants a
# writable cwd): a relative script path such as tools/sepbench.py keeps working.  Exit status: non-zero
# if any pass failed.
set -u
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
args=()
for a in "$@"; do
  if [[ "$a" != /* && "$a" != -* && -e "$root/$a" ]]; then args+=("$root/$a"); else args+=("$a"); fi
done
set -- "${args[@]}"
cd /tmp && export TMPDIR=/tmp
failed=0
passes=(
 "A:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
 "B:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "C:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
 "D:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "E:FETCH_SIZE"
 "F:WRITE_SIZE"
 "G:GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "H:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "I:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
 "J:SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL"
 "K:TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "L:TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "M:TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
 "O:TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
)
# PMC_PASSES="E F": only those passes (e.g. the two HBM-traffic counters)
for p in "${passes[@]}"; do
  name=${p%%:*}; ctrs=${p#*:}
  if [[ -n "${PMC_PASSES:-}" && " $PMC_PASSES " != *" $name "* ]]; then continue; fi
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1 || { echo "pass $name failed (see $out/$name.log)"; failed=1; }
done
echo "pmc passes done -> $out (failed=$failed)"
exit $failed
