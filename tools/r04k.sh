cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_refine.py -m gpu -q -s 2>&1 | grep -v Warning | grep "eval refine\|passed\|failed\|Error\|assert" | cut -c1-260 > gpurun_out/r04k/refine.txt; cat gpurun_out/r04k/refine.txt
( time python bench.py > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err ) 2>&1 | grep real; echo "bench rc=$?"; python - <<'PY'
import json
r=json.load(open("gpurun_out/r04k/bench.json"))
print({k:r[k] for k in ("value","ms_per_step","value_train")}, r["roofline"]["frac"], r["roofline"]["achieved"])
t=r["train"]; print({k:t[k] for k in ("ms_per_step","parity_1e3","max_abs_dlogp_vs_oracle_fp64","max_abs_dtrans_vs_oracle_fp64","tflops_executed")})
for k in ("fast_bf16x3","fast_bf16"): print(k, {a:b for a,b in t[k].items() if a!="mode" and a!="parity_note"})
print(r["config5"]); print(r["infer_fast_bf16x3"]["ms_per_step"], r["infer_fast_bf16"]["ms_per_step"])
PY
timeout 200 python tools/bench_pipeline.py 2>/dev/null | tail -3 | cut -c1-200
