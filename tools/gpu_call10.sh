set -x
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/prof
# phase profile of the four-wave Winograd kernel on the five layers that use it (zero data: the clock holds)
( cd tools/mb; for a in "16 240 320 256 256" "16 120 160 512 512" "16 60 80 1024 1024" "16 60 80 1024 512" "16 60 80 512 256"; do echo "== wino3_kernel N H W Cin Cout = $a"; timeout 60 ./w3_epi $a | grep -v "distinct"; timeout 60 ./w3_tl $a | grep segment; done ) > gpurun_out/prof/r02_wino3_phase_profile.log 2>&1
timeout 1500 bash tools/profile_round.sh r02 > gpurun_out/prof/profile_round.log 2>&1
tail -5 gpurun_out/prof/profile_round.log
head -c 600 gpurun_out/prof/r02_bench_final.json
