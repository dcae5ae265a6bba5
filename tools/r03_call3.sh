#!/bin/bash
O=gpurun_out/r03_c3
mkdir -p $O
bash tools/ab.sh $O/ab hip nocorr 2>&1 | tee $O/ab.txt
MODE=1 bash tools/ab.sh $O/ab_fast hip 2>&1 | tee -a $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_database.py -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log
timeout -k 10 300 python tools/fuzz_parity.py 80 41 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
