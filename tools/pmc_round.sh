# MFMA-busy counters for every kernel of the bench, and HBM bytes of the inference trunk in two SEPARATE
# single-counter passes (a combined FETCH_SIZE+WRITE_SIZE pass hangs on this image).  usage: bash tools/pmc_round.sh TAG [hbm]
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pmc2 /tmp/pmcF /tmp/pmcW
( cd /tmp && timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/pmc2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-epoch --min-seconds 0 > /tmp/p2.log 2>&1; echo "pmc2 rc=$?" )
DB=$(find /tmp/pmc2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_pmc_mfma.md "bench.py PMC pass, MFMA busy cycles / GUI active / LDS bank conflicts=$DB" > /dev/null
if [ "$2" = "hbm" ]; then
( cd /tmp && timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcF -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-fast --no-pmc --min-seconds 0 > /tmp/pf.log 2>&1; echo "pmcF rc=$?" )
DB=$(find /tmp/pmcF -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_pmc_fetch.md "bench.py PMC pass, FETCH_SIZE (KB; x2 on gfx950 for wide streaming reads)=$DB" > /dev/null
( cd /tmp && timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcW -o pw -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-fast --no-pmc --min-seconds 0 > /tmp/pw.log 2>&1; echo "pmcW rc=$?" )
DB=$(find /tmp/pmcW -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_pmc_write.md "bench.py PMC pass, WRITE_SIZE (KB)=$DB" > /dev/null
python tools/pmc_to_json.py gpurun_out/${TAG}_pmc_trunk.json $(find /tmp/pmc2 -name "*.db" | head -1) $(find /tmp/pmcF -name "*.db" | head -1) $(find /tmp/pmcW -name "*.db" | head -1) "$(cat .commit_id 2>/dev/null || echo unknown)" > /dev/null 2>&1; echo "pmc json rc=$?"
fi
ls -la gpurun_out | tail -5
