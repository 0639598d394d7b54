cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py --pmc > gpurun_out/r04_bench_final.json 2> /tmp/bench.err; echo "bench rc=$?"
bash tools/prof_round.sh r04 gpg > /tmp/prof.log 2>&1; grep "rc=" /tmp/prof.log
bash tools/pmc_round.sh r04 hbm > /tmp/pmc.log 2>&1; grep "rc=" /tmp/pmc.log
head -8 gpurun_out/r04_bench_trace.md; cat gpurun_out/r04_pmc_trunk.json | head -20
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04_bench_final.json")); print(r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["traffic"], r["config5"]["value"], r["train"]["ms_per_step"], r["train"]["fast_bf16x3"]["ms_per_step"])
PY
