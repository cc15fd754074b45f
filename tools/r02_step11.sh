cd /root/repo
timeout -k 10 900 python -m pytest tests/test_latent_distr.py tests/test_gpu_paris.py tests/test_gpu_kalman.py -x -q 2>&1 | tail -12
