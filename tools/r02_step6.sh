cd /root/repo
bash tools/ab_libs.sh r02e svm 3072 noshadow default
PFGRAD_LIB=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc/libpfgrad_stamps.so timeout -k 10 300 python tools/phase_profile.py svm 3072 > gpurun_out/r02e_phase_svm.txt 2>&1; cat gpurun_out/r02e_phase_svm.txt
timeout -k 10 600 python -m pytest tests/test_gpu_device_replay.py tests/test_gpu_ensemble.py -x -q 2>&1 | tail -4
