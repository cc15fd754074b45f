cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_device_replay.py tests/test_gpu_kalman.py tests/test_gpu_ensemble.py -x -q 2>&1 | tail -8
bash tools/ab_libs.sh r02h svm 3072 nobound default
bash tools/ab_libs.sh r02h_garch garch 4096 nobound default
