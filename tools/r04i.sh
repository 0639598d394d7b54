cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04i
timeout 300 python -m pytest tests/test_gpu_gpg.py -m gpu -q 2>&1 | tail -2
timeout 200 python tools/bench_gpg.py --P 3000 20000 50000 --cpu-draws 2 2>/dev/null | tail -3 > gpurun_out/r04i/bench_gpg.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r04i/bench_gpg.jsonl"):
    r=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ("P","gpu_s_per_scene","moments_kernel_ms","sweep_kernel_indexed_ms","index_build_ms","grasps")})
PY
PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_tm.so timeout 200 python tools/phase_times_x3.py 2>/dev/null > gpurun_out/r04i/phase_times_x3.txt; cat gpurun_out/r04i/phase_times_x3.txt
PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_tm.so timeout 200 python tools/phase_times.py 2>/dev/null | grep -v "^B " > gpurun_out/r04i/phase_times.txt
