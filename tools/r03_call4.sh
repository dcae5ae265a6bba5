#!/bin/bash
O=gpurun_out/r03_c4
mkdir -p $O
bash tools/ab.sh $O/ab hip nocorr corr1 corr2 nolicm 2>&1 | tee $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_database.py -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log
