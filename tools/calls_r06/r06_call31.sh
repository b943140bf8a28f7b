#!/bin/bash
# round 6, call 31: what bounds the PACKED first-layer backward-weight launch (99 us for 134 MB and 0.5 M MFMAs)?  Ablations of the packed
# instantiation, built from a patched COPY of the unit (the launcher's ablation switch names the unpacked instantiations)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06pa; mkdir -p $O
cd $R/tools/probe
sed 's/conv3d_wgrad_f16_pp_kernel<R, false, \([0-9]*\)>/conv3d_wgrad_f16_pp_kernel<R, (R == 32 || R == 16), \1>/' ../../pvcnn_amd/csrc/conv3d_wgrad_f16.hip > /tmp/conv3d_wgrad_f16_packed.hip
make api.o > $O/make.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-parameter -DPVCNN_ABLATE -I../../pvcnn_amd/csrc -I../../include -c /tmp/conv3d_wgrad_f16_packed.hip -o /tmp/ablate_packed.o >> $O/make.log 2>&1
cp libablate_conv3d_wgrad_f16.so /tmp/keep.so 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ablate_packed.o api.o -o libablate_conv3d_wgrad_f16.so >> $O/make.log 2>&1
cd $R
for ab in 0 2 4 16 1; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x9x64x32,16x9x64x16 2>>$O/err.log | tee -a $O/wgrad_packed_ablate.jsonl
done
cp /tmp/keep.so tools/probe/libablate_conv3d_wgrad_f16.so 2>/dev/null
tail -3 $O/make.log
