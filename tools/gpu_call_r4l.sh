cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
( echo "=== conv4b 60x80 1024->1024, 16 frames"; timeout 120 tools/mb/wino4_prof 16 60 80 1024 1024
  echo "=== conv3b 120x160 512->512, 16 frames"; timeout 120 tools/mb/wino4_prof 16 120 160 512 512
  echo "=== conv2b 240x320 256->256, 8 frames"; timeout 120 tools/mb/wino4_prof 8 240 320 256 256
  echo "=== conv1b 480x640 64->64, 8 frames"; timeout 120 tools/mb/wino4_prof 8 480 640 64 64 ) > gpurun_out/r4l/wino4_prof.log 2>&1
cat gpurun_out/r4l/wino4_prof.log
