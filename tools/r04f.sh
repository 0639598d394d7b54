cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04f
timeout 1200 python -m pytest tests/test_gpu_grad_gate.py tests/test_gpu_refine.py -m gpu -q -s 2>&1 | grep -v Warning | grep "gate B=\|refine\|passed\|failed\|Error" | cut -c1-300 > gpurun_out/r04f/gates.txt
timeout 120 ./examples/cabi_index_consumer > gpurun_out/r04f/index_consumer.txt 2>&1; echo "index consumer rc=$?"
LD_LIBRARY_PATH=/opt/rocm/lib/llvm/lib/clang/22/lib/linux:/opt/rocm/lib HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0 timeout 400 pointnetgpd_amd/csrc/build/asan/cabi_index_consumer_asan > gpurun_out/r04f/index_consumer_asan.txt 2>&1; echo "asan rc=$?"
timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-900 > gpurun_out/r04f/configs.jsonl
rm -rf /tmp/ptx; ( cd /tmp && TRACE_B=1024 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ptx -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 10 bf16x3 > /tmp/ttx.log 2>&1 ); DB=$(find /tmp/ptx -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/r04f/trace_bf16x3.md "10 steps bf16x3 B 1024=$DB" > /dev/null
tail -4 gpurun_out/r04f/gates.txt; cat gpurun_out/r04f/index_consumer.txt | grep gpg; tail -2 gpurun_out/r04f/index_consumer_asan.txt; head -8 gpurun_out/r04f/trace_bf16x3.md | tail -5
python - <<'PY'
import json
for l in open("gpurun_out/r04f/configs.jsonl"):
    try: r=json.loads(l)
    except Exception: continue
    print(r["config"][:28], {k:v for k,v in r.items() if k.endswith("_ms")})
PY
