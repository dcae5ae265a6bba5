#!/bin/bash
# One GPU-box call of the development loop: A/B timing of library variants on the bench workload, then the checks that
# guard exactness.  usage (on the box, from the repo root): bash tools/gpu_call.sh OUT_DIR "variant ..." [pytest args]
O=$1; V=$2; shift 2
mkdir -p $O
bash tools/ab.sh $O/ab $V 2>&1 | tee $O/ab.txt
timeout -k 10 900 python -m pytest "$@" -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
timeout -k 10 240 python tools/fuzz_parity.py 60 91 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
