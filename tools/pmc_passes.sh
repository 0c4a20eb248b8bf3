#!/bin/bash
# Collects rocprofv3 evidence for one bench configuration: a kernel-trace summary
# and separate PMC passes (HBM traffic, L2 hit rate, VALU utilisation).
# usage: tools/pmc_passes.sh <outdir> <bench args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py "${BENCH_ARGS[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err" || return 1
}
BENCH_ARGS=("$@")
run_pass trace --kernel-trace --stats || exit 1
run_pass pmc_fetch --pmc FETCH_SIZE || exit 1
run_pass pmc_write --pmc WRITE_SIZE || exit 1
run_pass pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run_pass pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY || exit 1
run_pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run_pass pmc_sq2 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run_pass pmc_l1 --pmc TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum || exit 1
echo passes done
