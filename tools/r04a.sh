# round 4, first GPU call: the new parity gates, RCCL at world 1, strong-scaling baseline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_head_train.py tests/test_gpu_grad_gate.py -m gpu -q -s -x 2>&1 | grep -v Warning | tail -60 > gpurun_out/r04a/gates.txt
timeout 600 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r04a/rccl_fused.txt
timeout 400 python tools/bench_strong.py > gpurun_out/r04a/strong.jsonl 2> gpurun_out/r04a/strong.err
tail -5 gpurun_out/r04a/gates.txt; tail -5 gpurun_out/r04a/rccl_fused.txt; cat gpurun_out/r04a/strong.jsonl | cut -c1-400
