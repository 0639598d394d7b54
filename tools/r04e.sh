cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04e
timeout 120 ./examples/cabi_index_consumer > gpurun_out/r04e/index_consumer.txt 2>&1; echo "index consumer rc=$?"
bash tools/asan_run.sh r04 > /dev/null 2>&1; cp gpurun_out/r04_asan.txt gpurun_out/r04e/; grep -c "exit code: 0" gpurun_out/r04_asan.txt; grep -i "ERROR: AddressSanitizer\|exit code" gpurun_out/r04_asan.txt | head
PNGPD_GATE_DIAG=1 timeout 1200 python -m pytest tests/test_gpu_grad_gate.py -m gpu -q -s 2>&1 | grep "gate B=" > gpurun_out/r04e/gate_diag.txt
timeout 300 python -m pytest tests/test_gpu_cabi_consumer.py -m gpu -q 2>&1 | tail -3
cat gpurun_out/r04e/index_consumer.txt
