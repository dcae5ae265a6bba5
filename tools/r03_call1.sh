#!/bin/bash
# round 3, GPU call 1: where k_vote's time goes (evidence for profiles/r03_*)
O=gpurun_out/r03_c1
mkdir -p $O
./gpurun_ab/vote_mix > $O/vote_mix.txt 2>&1; cat $O/vote_mix.txt
bash tools/ab.sh $O/ab hip prof noloop noatom 2>&1 | tee $O/ab.txt
bash tools/pmc_sq.sh $O/pmc_sq > $O/pmc_sq.txt 2>&1; tail -60 $O/pmc_sq.txt
