# Final evidence of the round (GPU box): full GPU suite, smoke, default bench, per-config / scoring / sampler
# benches, kernel traces.  Every profiler call is bounded by its own timeout.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r01_bench_final.json 2> /tmp/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r01_bench_final.json
timeout 300 python tools/bench_configs.py > gpurun_out/r01_bench_configs.jsonl 2> /tmp/cfg.err; echo "cfg rc=$?"
timeout 120 python tools/bench_latency.py 2>/dev/null | tail -1 > gpurun_out/r01_bench_latency.json
timeout 200 python tools/bench_gpg.py --P 3000 20000 50000 --cpu-draws 2 2>/dev/null | tail -3 > gpurun_out/r01_bench_gpg.jsonl
timeout 200 python tools/bench_scoring.py 2>/dev/null | tail -1 > gpurun_out/r01_bench_scoring.json
timeout 200 python tools/bench_pipeline.py 2>/dev/null | tail -3 > gpurun_out/r01_bench_pipeline.jsonl
bash tools/prof_round.sh > /tmp/prof.log 2>&1; grep "rc=" /tmp/prof.log
ls gpurun_out
