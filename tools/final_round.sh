# Final evidence of the round (GPU box): full GPU suite, smoke, default bench, per-config bench, kernel traces.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r01_bench_final.json 2> /tmp/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r01_bench_final.json
timeout 300 python tools/bench_configs.py > gpurun_out/r01_bench_configs.jsonl 2> /tmp/cfg.err; echo "cfg rc=$?"
timeout 120 python tools/bench_latency.py 2>/dev/null | tail -1 > gpurun_out/r01_bench_latency.json
timeout 200 python tools/bench_gpg.py --cpu-draws 2 2>/dev/null | tail -2 > gpurun_out/r01_bench_gpg.jsonl
bash tools/prof_round.sh > /tmp/prof.log 2>&1; grep "rc=" /tmp/prof.log
( cd /tmp && B=64 N=750 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pst -o st -- python $GRAFT_REPO_ROOT/tools/trace_small_train.py > /tmp/st.log 2>&1 )
python tools/rocprof_summary.py gpurun_out/r01_final2_small_batch_train_trace.md "tools/trace_small_train.py, B 64 N 750, 12 eager steps=$(find /tmp/pst -name '*.db' | head -1)" > /dev/null
ls gpurun_out
