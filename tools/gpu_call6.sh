export TMPDIR=/tmp
for D in 0 1 2 3; do echo "== KFN_WINO2_DBG=$D"; KFN_WINO2_DBG=$D MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv4b python tools/mb_wino.py 2>&1 | grep FUSED | cut -c1-70; done
