#!/bin/bash
# The measurements that the documents quote, in one GPU call:  tools/round_measure.sh <tag>
#   gpurun_out/<tag>_bench_{prove,msm,ntt}.json   the three bench lines (prove with its CPU baseline leg)
#   gpurun_out/<tag>_bench_prove_8n.json          the proof forced over the reference's 8n coset, for comparison
#   gpurun_out/<tag>_small_proofs.txt             ms per proof at 2^10 .. 2^18 gates
# Every step has its own time limit and the chain stops at the first failure.
set -e
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
timeout -k 10 400 python3 bench.py > $out/${tag}_bench_prove.json 2> $out/${tag}_bench_prove.err
echo "prove line done"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-sub-records --full-coset > $out/${tag}_bench_prove_8n.json 2> $out/${tag}_bench_prove_8n.err
echo "prove over 8n done"
timeout -k 10 300 python3 bench.py --workload msm > $out/${tag}_bench_msm.json 2> $out/${tag}_bench_msm.err
echo "msm line done"
timeout -k 10 300 python3 bench.py --workload ntt > $out/${tag}_bench_ntt.json 2> $out/${tag}_bench_ntt.err
echo "ntt line done"
: > $out/${tag}_small_proofs.txt
for lg in 10 12 13 14 15 16 17 18; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-sub-records --steps 20 --log-gates $lg > $out/small.json 2> $out/small.err
  python3 -c "
import json
d = json.load(open('$out/small.json'))
print('2^%d gates: %.3f ms per proof, coset %s' % ($lg, d['ms_per_step'], d.get('quotient_coset', '')[:9]))" >> $out/${tag}_small_proofs.txt
done
rm -f $out/small.json $out/small.err
cat $out/${tag}_small_proofs.txt
