#!/bin/bash
# round 3, GPU call 2: search-based exact mode (A'), new multi-rank and database tests
O=gpurun_out/r03_c2
mkdir -p $O
bash tools/ab.sh $O/ab hip 2>&1 | tee $O/ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -40 $O/pytest.log
timeout -k 10 400 python tools/fuzz_parity.py 120 31 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
