#!/bin/bash
# Bench lines, kernel-trace summaries, the proof's timeline and the measured CPU baseline in one GPU call (one
# box: the times belong together):  tools/round_profiles.sh <tag>   -> gpurun_out/<tag>_*
set -e
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
tools/round_measure.sh $tag
tools/profile_all.sh
for w in prove msm ntt; do
  cp $(ls gpurun_out/prof_$w/*kernel_stats.csv | head -1) gpurun_out/${tag}_${w}_kernel_stats.csv
  cp gpurun_out/bench_${w}_under_rocprof.json gpurun_out/${tag}_bench_${w}_under_rocprof.json
done
# a key in which every widget has gates: kernel-trace summary of tools/widget_quotient_ab.py (key load + 6 proofs)
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_widgets -o widgets -- \
  python3 $root/tools/widget_quotient_ab.py > $root/gpurun_out/${tag}_widgets_under_rocprof.txt 2> $root/gpurun_out/prof_widgets.log)
cp $(ls gpurun_out/prof_widgets/*kernel_stats.csv | head -1) gpurun_out/${tag}_widgets_kernel_stats.csv
echo "widget key done"
python3 tools/timeline_gaps.py $(ls gpurun_out/prof_prove/*kernel_trace.csv | head -1) > gpurun_out/${tag}_prove_timeline.txt
echo "timeline done"
timeout -k 10 900 python3 bench.py --cpu-baseline-full --no-sub-records > gpurun_out/${tag}_cpu_baseline_prove.json 2> gpurun_out/${tag}_cpu_baseline_prove.err
echo "cpu baseline done"
