#!/bin/bash
# Hardware-counter passes for one bench workload, each in its own rocprofv3 run (counters of one block do not split
# over passes).   usage: tools/pmc.sh <tag> <bench args...>   -> gpurun_out/pmc_<tag>_{fetch,write,sq1,sq2}/
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters
  timeout -s KILL 240 rocprofv3 --pmc $2 --output-format csv -d $root/gpurun_out/pmc_${tag}_$1 -o pmc -- python3 $root/bench.py "${@:3}" > $root/gpurun_out/pmc_${tag}_$1.log 2>&1
  echo "pass $1 done"
}
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "$@"
run sq2 "SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" "$@"
