#!/bin/bash
# Debugging builds of the library for tools/debug_conv64.py (MB_LIB=<path>): C64_PADS="-1 0 1" -> tools/mb/libkfnet_pad<p>.so with
# conv64_rows_kernel compiled with KFN_STORE_PAD=<p> (kfn_common.h buffer_store_b128: -1 = no wait states behind a 16-byte buffer
# store = round 4's kernel; p >= 0 = s_nop p).
cd "$(dirname "$0")/../.."
OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v kfn_conv64.o)
for v in ${C64_PADS--1 0 1}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_STORE_PAD=$v -c kfnet_amd/csrc/kfn_conv64.hip -o /tmp/kfn_conv64_pad$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_pad$v.so /tmp/kfn_conv64_pad$v.o $OBJS -lz ) &
done
wait
ls -la tools/mb/libkfnet_pad*.so
