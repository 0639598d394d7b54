cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04j
timeout 200 python tools/bench_gpg.py --P 3000 20000 50000 --cpu-draws 2 2>/dev/null | tail -3 > gpurun_out/r04j/bench_gpg.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r04j/bench_gpg.jsonl"):
    r=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ("P","gpu_s_per_scene","moments_kernel_ms","sweep_kernel_indexed_ms","index_build_ms","grasps")})
PY
timeout 200 python tools/bench_pipeline.py 2>/dev/null | tail -3 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r04j/suite.txt; cat gpurun_out/r04j/suite.txt
