# PMC passes over the kernels of a TRAINING step (tools/trace_train.py, B = N = 1024 fp32), each counter group in its own
# run (FETCH_SIZE and WRITE_SIZE never share a pass; no trace domain besides --kernel-trace).
#   usage: bash tools/pmc_train.sh TAG [fp32|bf16x3|bf16]
#   -> gpurun_out/TAG_pmc_wait_train.md   wave cycles / parked / issue-stalled / issuing (quad-cycles)
#      gpurun_out/TAG_pmc_fetch_train.md  FETCH_SIZE (KB; x2 on gfx950 for wide streaming reads)
#      gpurun_out/TAG_pmc_write_train.md  WRITE_SIZE (KB)
#      gpurun_out/TAG_pmc_mfma_train.md   MFMA busy cycles / GUI active / LDS conflicts
TAG=${1:-r03}
PREC=${2:-fp32}
SUF=""; [ "$PREC" != "fp32" ] && SUF="_$PREC"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run_pass() {   # name, label, counters...
  local name=$1 label=$2; shift 2
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 240 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/tools/trace_train.py 3 $PREC > /tmp/pmc_$name.log 2>&1; echo "pmc $name rc=$?" )
  local DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_pmc_${name}_train${SUF}.md "3 training steps at B 1024 N 1024 $PREC, $label=$DB" > /dev/null
}
run_pass wait "SQ wave cycles / parked (WAIT_ANY) / issue-stalled (WAIT_INST_ANY) / issuing (ACTIVE_INST_*), quad-cycles" \
  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run_pass fetch "FETCH_SIZE (KB; x2 on gfx950 for wide streaming reads)" FETCH_SIZE
run_pass write "WRITE_SIZE (KB)" WRITE_SIZE
run_pass mfma "MFMA busy cycles / GUI active / LDS bank conflicts / LDS active" SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
ls -la gpurun_out | grep ${TAG}_pmc
