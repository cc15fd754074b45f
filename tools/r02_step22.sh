cd /root/repo
bash tools/ab_libs.sh r02j svm 3072 default early1 early2 top topearly2
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
PFGRAD_LIB=$CS/libpfgrad_topearly2.so timeout -k 10 300 python -m pytest tests/test_gpu_device_replay.py -x -q 2>&1 | tail -3
