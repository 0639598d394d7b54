cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fused.py tests/test_gpu_train_large.py -x -q 2>&1 | tail -3
bash tools/ab_lib.sh bwd_d 128 1024
