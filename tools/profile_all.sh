#!/bin/bash
# Kernel-trace summaries of the three bench workloads, one rocprofv3 run each:
#   tools/profile_all.sh  ->  gpurun_out/prof_<workload>/ (rocprofv3 CSVs), gpurun_out/bench_<workload>_under_rocprof.json
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for w in ${@:-prove msm ntt}; do
  # the prove line's msm / ntt sub-records would mix their kernels into the proof's summary: they get runs of their own
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$w -o $w -- \
    python3 $root/bench.py --workload $w --no-cpu-baseline --no-sub-records > $root/gpurun_out/bench_${w}_under_rocprof.json 2> $root/gpurun_out/prof_$w.log
  echo "kernel trace of $w done"
done
