#!/bin/bash
# Several processes with several contexts each on one GPU, timed over a common window:
#   tools/concurrent_processes.sh LOG_GATES PROCESSES THREADS_PER_PROCESS [PROOFS_PER_THREAD]
# A process has four hardware queues; more processes give more kernels in flight at once.  The rates printed by the
# processes add up (the windows start together and end within a few percent of each other).
lg=$1; procs=$2; threads=$3; per=${4:-150}
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import torch" > /dev/null 2>&1   # the first import on a fresh machine takes a minute or two
start=$(python3 -c "import time; print(time.time() + 30)")
pids=()
for p in $(seq 1 $procs); do
  timeout -k 10 200 python3 $root/tools/concurrent_proofs.py $lg $threads $per $start > $root/gpurun_out/conc_p$p.log 2>&1 &
  pids+=($!)
done
rc=0
for pid in "${pids[@]}"; do wait $pid || rc=1; done
cat $root/gpurun_out/conc_p*.log | grep "proofs/s"
python3 - <<PY
import glob, re
tot = 0.0
for f in glob.glob("$root/gpurun_out/conc_p*.log"):
    m = re.search(r"-> ([0-9.]+) proofs/s", open(f).read())
    if m: tot += float(m.group(1))
print("2^$lg gates, $procs processes x $threads contexts: %.0f proofs/s in total" % tot)
PY
rm -f $root/gpurun_out/conc_p*.log
exit $rc
