#!/bin/bash
# The secondary evidence of a round in one gpurun call (the bench line, kernel stats and PMC traffic come from
# tools/profile_round.sh):  tools/round_evidence.sh r03     (run on the GPU box from the repo root)
R=${1:-r03}
O=gpurun_out/${R}_evidence
mkdir -p $O
for c in cfg2 cfg3 cfg3db cfg4 cfg5 db50 planes; do timeout -k 10 300 python tools/bench_configs.py $c 2> $O/$c.err | tail -1 >> $O/configs.jsonl; echo "done $c"; done
timeout -k 10 300 python tools/host_tail_bound.py 2> $O/host_tail.err | grep '^{' > $O/host_tail.jsonl; echo "done host tail"
timeout -k 10 600 python tools/recall_occlusion.py 8 > $O/recall_occlusion.json 2> $O/recall.err; echo "done recall"
timeout -k 10 500 python tools/fuzz_parity.py 200 21 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
