#!/bin/bash
# SQ counter passes over the bench (vote kernel): tools/pmc_sq.sh out_dir   (run on the GPU box from the repo root)
# Separate rocprofv3 --pmc runs, no tracing alongside (see the profiling rules of the pool).
O=${1:-gpurun_out/pmc_sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES"; do
  rocprofv3 --output-format csv --pmc $set -d $O/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/p$i.json 2> $O/p$i.err || echo "pass $i failed: $(tail -2 $O/p$i.err)"
  i=$((i+1))
done
python3 tools/pmc_summary.py $O k_vote > $O/summary_k_vote.txt
cat $O/summary_k_vote.txt
