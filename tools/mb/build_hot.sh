#!/bin/bash
# Timing-experiment builds of the conv kernels with "hot" operands (NOT product builds, results are wrong on purpose):
#   tools/mb/libkfnet_hot1.so  every A (activation) load of the fp16-activation kernels hits a 64 KiB window
#   tools/mb/libkfnet_hot2.so  every B (weight) load re-reads the first K chunk
#   tools/mb/libkfnet_hot3.so  both
# They answer "is the kernel bound by where its operands come from?" (tools/mb_f16.py with MB_LIB=<path>).
cd "$(dirname "$0")/../.."
for v in ${HOT_VARIANTS-1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_CONV_HOT=$v -c kfnet_amd/csrc/kfn_conv.hip -o /tmp/kfn_conv_hot$v.o || exit 1
  OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v kfn_conv.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_hot$v.so /tmp/kfn_conv_hot$v.o $OBJS -lz || exit 1
done
ls -la tools/mb/*.so
# A/B of wino2_kernel's two epilogues on ALIGNED outputs: tools/mb/libkfnet_w2dword.so forces the dword-store form
if [ -n "$W2_DWORD" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_WINO2_NO_WIDE -c kfnet_amd/csrc/kfn_wino2.hip -o /tmp/kfn_wino2_nw.o || exit 1
  OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v kfn_wino2.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_w2dword.so /tmp/kfn_wino2_nw.o $OBJS -lz || exit 1
fi
# A/B of non-temporal wide output stores in the convolution kernels: tools/mb/libkfnet_ntstore.so
if [ -n "$NT_STORE" ]; then
  OB=""
  for f in kfn_wino3 kfn_wino2 kfn_wino_s2 kfn_conv; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DKFN_NT_STORE_AUX=2 -c kfnet_amd/csrc/$f.hip -o /tmp/${f}_nt.o || exit 1
    OB="$OB /tmp/${f}_nt.o"
  done
  OBJS=$(ls kfnet_amd/csrc/build/*.o | grep -v "kfn_wino3.o\|kfn_wino2.o\|kfn_wino_s2.o\|kfn_conv.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/libkfnet_ntstore.so $OB $OBJS -lz || exit 1
fi
