cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04c
PNGPD_GATE_DIAG=1 timeout 900 python -m pytest tests/test_gpu_grad_gate.py -m gpu -q -s 2>&1 | grep "gate B=" > gpurun_out/r04c/gate_diag.txt
timeout 300 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_rccl.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r04c/ddp.txt
timeout 300 python tools/bench_strong.py 2>/dev/null | cut -c1-300 > gpurun_out/r04c/strong.jsonl
cat gpurun_out/r04c/ddp.txt; cat gpurun_out/r04c/strong.jsonl | head -6
