cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
( for t in a b c d e; do echo "=== variant $t (a: timeline of super-step 20, NB 9; b: NB 18; c: super-step 21; d: no patch loads; e: L2-hot weights) conv4b 60x80 1024->1024, 16 frames"; timeout 120 tools/mb/wino4_prof_$t 16 60 80 1024 1024 | tail -4; done ) > gpurun_out/r4m/wino4_prof.log 2>&1
cat gpurun_out/r4m/wino4_prof.log
