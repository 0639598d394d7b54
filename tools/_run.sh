cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fused.py tests/test_gpu_train_large.py tests/test_gpu_grad_gate.py -x -q 2>&1 | tail -3
rm -rf /tmp/ptq; ( cd /tmp && TRACE_B=128 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ptq -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 20 fp32 > /tmp/tr.log 2>&1 )
DB=$(find /tmp/ptq -name "*.db" | head -1)
python tools/rocprof_summary.py --all gpurun_out/r04s_trace_B128.md "20 eager training steps at B 128 N 1024 fp32 (tools/trace_train.py)=$DB" > /dev/null
grep -i "dw3_fin\|a_cvec\|trunk_bwd_e\|trunk_bwd_d" gpurun_out/r04s_trace_B128.md
bash tools/ab_lib.sh none 128 1024 | tail -8
