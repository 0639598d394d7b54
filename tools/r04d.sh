cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04d
PNGPD_GATE_DIAG=1 timeout 900 python -m pytest tests/test_gpu_grad_gate.py -m gpu -q -s 2>&1 | grep "gate B=" > gpurun_out/r04d/gate_diag.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_large.py tests/test_gpu_fused.py tests/test_gpu_bf16.py tests/test_gpu_refine.py tests/test_gpu_cabi_consumer.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04d/suite.txt
timeout 300 python tools/bench_strong.py --no-rccl 2>/dev/null | cut -c1-200 > gpurun_out/r04d/strong.jsonl
for B in 128 1024; do rm -rf /tmp/pt$B; ( cd /tmp && TRACE_B=$B timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt$B -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 20 fp32 > /tmp/tt$B.log 2>&1 ); DB=$(find /tmp/pt$B -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/r04d/trace_B$B.md "20 steps B $B N 1024=$DB" > /dev/null; done
rm -rf /tmp/ptx; ( cd /tmp && TRACE_B=1024 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ptx -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 10 bf16x3 > /tmp/ttx.log 2>&1 ); DB=$(find /tmp/ptx -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/r04d/trace_bf16x3.md "10 steps bf16x3 B 1024=$DB" > /dev/null
cat gpurun_out/r04d/suite.txt; cat gpurun_out/r04d/strong.jsonl | head -5
