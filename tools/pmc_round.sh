#!/bin/bash
# All hardware-counter passes that the documents quote, in one GPU call:  tools/pmc_round.sh <tag>
#   gpurun_out/<tag>_pmc_{ntt,msm20,msm24,prove,prove_8n,widgets}.csv   (tools/pmc.sh passes, summarised by tools/pmc_summary.py)
set -e
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
common="--no-cpu-baseline --no-sub-records --steps 3 --warmup 1"
tools/pmc.sh ntt --workload ntt $common
tools/pmc.sh msm20 --workload msm --log-points 20 $common
tools/pmc.sh msm24 --workload msm --log-points 24 $common
tools/pmc.sh prove $common
tools/pmc.sh prove_8n $common --full-coset
tools/pmc.sh widgets $common --circuit widgets
for t in ntt msm20 msm24 prove prove_8n widgets; do
  python3 tools/pmc_summary.py gpurun_out/pmc_$t > gpurun_out/${tag}_pmc_$t.csv
  echo "$t: $(wc -l < gpurun_out/${tag}_pmc_$t.csv) rows"
done
