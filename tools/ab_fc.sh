#!/bin/bash
# A/B of the product library against pointnetgpd_amd/csrc/build/variants/lib_base.so, alternating processes
cd /root/repo
O=gpurun_out/ab_fc; mkdir -p $O; : > $O/ab.jsonl
timeout 300 python -m pytest tests/test_gpu_head_train.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
for r in 1 2; do
  timeout 300 python tools/ab_fc.py "$@" >> $O/ab.jsonl 2>/dev/null
  PNGPD_LIB=$GRAFT_REPO_ROOT/pointnetgpd_amd/csrc/build/variants/lib_base.so timeout 300 python tools/ab_fc.py "$@" >> $O/ab.jsonl 2>/dev/null
done
cat $O/ab.jsonl
