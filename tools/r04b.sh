cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_gpu_refine.py -m gpu -q -s 2>&1 | grep -v Warning | tail -40 > gpurun_out/r04b/refine.txt
timeout 900 python -m pytest tests/test_gpu_head_train.py tests/test_gpu_grad_gate.py -m gpu -q -s 2>&1 | grep -v Warning | tail -60 > gpurun_out/r04b/gates.txt
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fused.py tests/test_gpu_infer_x3.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r04b/bf16.txt
for B in 128 1024; do rm -rf /tmp/pt$B; ( cd /tmp && TRACE_B=$B timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt$B -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 20 fp32 > /tmp/tt$B.log 2>&1 ); DB=$(find /tmp/pt$B -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/r04b/trace_B$B.md "20 steps B=$B N=1024=$DB" > /dev/null; done
rm -rf /tmp/ptx; ( cd /tmp && TRACE_B=1024 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ptx -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 10 bf16x3 > /tmp/ttx.log 2>&1 ); DB=$(find /tmp/ptx -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/r04b/trace_bf16x3.md "10 steps bf16x3 B=1024=$DB" > /dev/null
tail -3 gpurun_out/r04b/refine.txt; tail -3 gpurun_out/r04b/gates.txt; tail -3 gpurun_out/r04b/bf16.txt
