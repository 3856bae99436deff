#!/bin/bash
# rebuild libavbd_hip.so with the in-kernel time stamps (tools/README.md); `make -B -C avbd-demo3d_amd/csrc` goes back to the product build
make -B -C /root/repo/avbd-demo3d_amd/csrc HIPFLAGS="-std=c++17 -O2 --offload-arch=gfx950 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-fast-math -DAVBD_DF_STATS $*" 2>&1 | grep -E "error|Error" | head
