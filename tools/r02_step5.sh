cd /root/repo
bash tools/ab_libs.sh r02d svm 3072 xoshiro default
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
