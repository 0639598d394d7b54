# usage: bash tools/trace_train.sh TAG [fp32|bf16x3|bf16]  -> gpurun_out/TAG_train_step_trace[_PREC].md (10 steps, every kernel listed)
TAG=${1:-r02}
PREC=${2:-fp32}
SUF=""; [ "$PREC" != "fp32" ] && SUF="_$PREC"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/prof_tr
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/tools/trace_train.py 10 $PREC > /tmp/tr.log 2>&1; echo "tr rc=$?" )
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/${TAG}_train_step_trace${SUF}.md "10 eager training steps at B 1024 N 1024 k 2, precision $PREC (tools/trace_train.py)=$DB" > /dev/null
