#!/bin/bash
# Timing-experiment builds of wino4_kernel (NOT product builds, results are wrong on purpose): KFN_W4_DBG bit 0 = no input
# transform, 1 = no patch loads, 2 = no V stores, 3 = no B loads in the main loop.  tools/mb_wino.py with MB_LIB=<path>.
#   W4_VARIANTS="1 2 8 15" tools/mb/build_w4.sh      W4_DEFS="-DKFN_W4_XSLOT=90 ..." W4_TAG=x90 tools/mb/build_w4.sh
cd "$(dirname "$0")/../.."
OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v kfn_wino4.o)
for v in ${W4_VARIANTS-1 2 4 8 15}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_W4_DBG=$v -c kfnet_amd/csrc/kfn_wino4.hip -o /tmp/kfn_wino4_dbg$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_w4dbg$v.so /tmp/kfn_wino4_dbg$v.o $OBJS -lz ) &
done
if [ -n "$W4_DEFS" ]; then
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $W4_DEFS -c kfnet_amd/csrc/kfn_wino4.hip -o /tmp/kfn_wino4_$W4_TAG.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_w4$W4_TAG.so /tmp/kfn_wino4_$W4_TAG.o $OBJS -lz ) &
fi
wait
ls -la tools/mb/libkfnet_w4*.so
