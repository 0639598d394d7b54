cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04g
V=$GRAFT_REPO_ROOT/pointnetgpd_amd/csrc/build/variants/lib_x3lock.so
for i in 1 2; do
  echo "== pipelined (product)"; timeout 120 python tools/bench_eval_bf.py 2>/dev/null
  echo "== lock-step (variant)"; PNGPD_LIB=$V timeout 120 python tools/bench_eval_bf.py 2>/dev/null
done > gpurun_out/r04g/x3_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_infer_x3.py tests/test_gpu_bf16.py tests/test_gpu_infer.py tests/test_gpu_fused.py tests/test_gpu_train.py tests/test_gpu_gpd.py tests/test_gpu_ddp.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r04g/suite.txt
timeout 300 python tools/bench_strong.py --no-rccl 2>/dev/null | cut -c1-200 > gpurun_out/r04g/strong.jsonl
timeout 120 python tools/bench_step.py 2>/dev/null > gpurun_out/r04g/step.txt
cat gpurun_out/r04g/x3_ab.txt; cat gpurun_out/r04g/suite.txt; head -5 gpurun_out/r04g/strong.jsonl; cat gpurun_out/r04g/step.txt
