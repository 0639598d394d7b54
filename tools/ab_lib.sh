#!/bin/bash
# A/B of the product library against pointnetgpd_amd/csrc/build/variants/lib_base.so (alternating processes):
# per-pass times (tools/bench_pass.py) and the whole fp32 step (tools/ab_fc.py).  usage: ab_lib.sh <passes> [B ...]
cd /root/repo
PASSES=$1; shift
O=gpurun_out/ab_lib; mkdir -p $O; : > $O/pass.txt; : > $O/step.jsonl
V=$GRAFT_REPO_ROOT/pointnetgpd_amd/csrc/build/variants/lib_base.so
for r in 1 2; do for B in "$@"; do
  echo "== product B $B" >> $O/pass.txt; PNGPD_BENCH_B=$B PNGPD_PASSES=$PASSES timeout 120 python tools/bench_pass.py 2>/dev/null | grep -v "^B " >> $O/pass.txt
  echo "== base B $B" >> $O/pass.txt; PNGPD_LIB=$V PNGPD_BENCH_B=$B PNGPD_PASSES=$PASSES timeout 120 python tools/bench_pass.py 2>/dev/null | grep -v "^B " >> $O/pass.txt
done
timeout 300 python tools/ab_fc.py "$@" 2>/dev/null >> $O/step.jsonl
PNGPD_LIB=$V timeout 300 python tools/ab_fc.py "$@" 2>/dev/null >> $O/step.jsonl
done
cat $O/pass.txt
python - <<'PY'
import json
for ln in open("gpurun_out/ab_lib/step.jsonl"):
    d = json.loads(ln); print(d["lib"], d["B"], d["step_eager_ms"], d["step_graph_ms"])
PY
