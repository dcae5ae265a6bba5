#!/bin/bash
# Everything a round's profiles/ entries come from, in one gpurun call:
#   tools/profile_round.sh r03     (run on the GPU box from the repo root)
# bench line, rocprofv3 kernel stats of the same command, and the two PMC traffic passes.
set -e
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kstats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/kstats.json 2> $O/kstats.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
python3 tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json      # bench.py reads the traffic figure from here
python3 bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
python3 bench.py --steps 5 --warmup 1 --vote-mode fast --no-cpu-baseline > $O/bench_n1_fast.json 2> $O/bench_n1_fast.err
cp $O/kstats/*/*kernel_stats.csv $O/kernel_stats.csv
cat $O/bench_n1.json
