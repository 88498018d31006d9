set -x
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/prof
( cd tools/mb; for a in "16 240 320 256 256" "16 120 160 512 512" "16 60 80 1024 1024" "16 60 80 1024 512" "16 60 80 512 256"; do echo "== wino3_kernel N H W Cin Cout = $a"; timeout 60 ./w3_epi $a | grep -v "distinct"; timeout 60 ./w3_tl $a | grep segment; done ) 2>&1 | grep -v "^+" > gpurun_out/prof/r02_wino3_phase_profile.log
( MB_F16=1 MB_FUSED_ONLY=1 MB_BATCH=16 python tools/mb_wino.py; MB_F16=1 python tools/mb_s2.py; MB_FUSED_ONLY=1 MB_BATCH=16 python tools/mb_wino.py; python tools/mb_s2.py ) 2>&1 | grep -v "amdgpu\|^+" > gpurun_out/prof/r02_layer_microbench.log
timeout 1500 bash tools/profile_round.sh r02 > gpurun_out/prof/profile_round.log 2>&1
timeout 300 python bench.py --config c2 > gpurun_out/prof/r02_bench_c2.json 2> gpurun_out/prof/c2.err
timeout 400 python bench.py --config c5 > gpurun_out/prof/r02_bench_c5.json 2> gpurun_out/prof/c5.err
timeout 1200 python -m pytest tests -m gpu -q -W ignore > gpurun_out/prof/r02_gpu_tests_final.log 2>&1
tail -3 gpurun_out/prof/r02_gpu_tests_final.log
head -c 400 gpurun_out/prof/r02_bench_final.json; echo
cat gpurun_out/prof/r02_layer_microbench.log | cut -c1-130
