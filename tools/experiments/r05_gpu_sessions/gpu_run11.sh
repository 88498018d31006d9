#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( for u in 0 1; do for a in "256 64" "256 256" "256 32 68 120" "4 64 68 120"; do echo "=== KFN_SCAN_UNIFORM_SRD=$u  args $a"; timeout 300 tools/mb/kalman_mb_srd$u $a | grep -v "^fuse\|^# kalman fuse"; done; done ) > gpurun_out/r05_kalman_srd_ab.log 2>&1
cat gpurun_out/r05_kalman_srd_ab.log; grep -c DIFFERS gpurun_out/r05_kalman_srd_ab.log
