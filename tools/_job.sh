exec < /dev/null
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_trained.py -m gpu -q -s > gpurun_out/r2_trained.log 2>&1; grep -E "cfg-5 on|passed|failed|Wds:|solver iterations|rounding-stable|device \{|^E " gpurun_out/r2_trained.log | cut -c1-300
