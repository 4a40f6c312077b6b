#!/bin/bash
# The randomized soaks of tests/soak back to back on a GPU box; one line per soak in gpurun_out/soaks.log
mkdir -p gpurun_out
log=gpurun_out/soaks.log
: > $log
run() { local t0=$(date +%s); timeout 900 python "$@" > gpurun_out/soak_last.txt 2>&1; local rc=$?; echo "$* -> rc $rc ($(( $(date +%s) - t0 )) s): $(tail -1 gpurun_out/soak_last.txt)" >> $log; }
run tests/soak/soak_parity.py ${1:-60}
run tests/soak/soak_grid64.py 30
run tests/soak/soak_large.py 1
run tests/soak/soak_large.py 14
run tests/soak/soak_large.py 40
run tests/soak/soak_huge.py
run tests/soak/soak_observe.py 24
run tests/soak/soak_pipeline.py 2
cat $log
