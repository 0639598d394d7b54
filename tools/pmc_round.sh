# MFMA-busy counters for every kernel of the bench (bounded: a hung profiler costs at most the timeout).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && timeout 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/pmc2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/p2.log 2>&1; echo "pmc2 rc=$?" )
DB=$(find /tmp/pmc2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/r01_final2_pmc_mfma.md "bench.py PMC pass, MFMA busy cycles / GUI active / LDS bank conflicts=$DB" > /dev/null
( cd /tmp && timeout 170 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/pmc1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train > /tmp/p1.log 2>&1; echo "pmc1 rc=$?" )
DB=$(find /tmp/pmc1 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/r01_final2_pmc_hbm.md "bench.py PMC pass, FETCH_SIZE / WRITE_SIZE (KB; FETCH_SIZE x2 on gfx950)=$DB" > /dev/null
ls -la gpurun_out
