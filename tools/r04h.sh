cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04h
timeout 600 python -m pytest tests/test_gpu_gpg.py tests/test_gpu_crop_scoring.py -m gpu -q 2>&1 | tail -4 > gpurun_out/r04h/gpg_tests.txt
timeout 200 python tools/bench_gpg.py --P 3000 20000 50000 --cpu-draws 2 2>/dev/null | tail -3 > gpurun_out/r04h/bench_gpg.jsonl
timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "sgd_steps" 2>&1 | grep -v Warning | tail -25 > gpurun_out/r04h/sgd.txt
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_infer_x3.py tests/test_gpu_bf16.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r04h/fused.txt
cat gpurun_out/r04h/gpg_tests.txt; python - <<'PY'
import json
for l in open("gpurun_out/r04h/bench_gpg.jsonl"):
    r=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ("P","gpu_s_per_scene","moments_kernel_ms","sweep_kernel_indexed_ms","index_build_ms","grasps")})
PY
grep -n "losses\|eval max\|assert\|Error\|passed\|failed" gpurun_out/r04h/sgd.txt | head; cat gpurun_out/r04h/fused.txt
