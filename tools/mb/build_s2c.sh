#!/bin/bash
# Timing-experiment builds of wino_s2c_kernel (NOT product builds: results are wrong on purpose): tools/mb/libkfnet_s2c_<mask>.so with
# KFN_S2C_EXP=<mask> (bit 0 no mid barrier, 1 no gathers, 2 no transform, 3 no V stores, 4 no weight loads in the loop, 5 no V reads in
# the loop).  Use: S2C_VARIANTS="1 14 16 32" bash tools/mb/build_s2c.sh; MB_LIB=tools/mb/libkfnet_s2c_14.so MB_S2_FORM=5 python tools/mb_s2.py
cd "$(dirname "$0")/../.."
OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v kfn_wino_s2c.o)
for v in ${S2C_VARIANTS-1 14 16 32}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_S2C_EXP=$v ${S2C_DEFS} -c kfnet_amd/csrc/kfn_wino_s2c.hip -o /tmp/kfn_s2c_$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_s2c_$v.so /tmp/kfn_s2c_$v.o $OBJS -lz ) &
done
wait
ls -la tools/mb/libkfnet_s2c_*.so
