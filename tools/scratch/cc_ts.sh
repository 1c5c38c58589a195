#!/bin/bash
# compiles one csrc TU to ISA and prints per-kernel register use: tools/scratch/cc_ts.sh train_stream_kernels.hip
cd /root/repo/point-cloud-reid_amd/csrc
mkdir -p /tmp/t
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. -S --cuda-device-only $1 -o /tmp/t/${1%.hip}.s 2>&1 | grep -E "error|warning: v" -A5
grep -E "^_ZN.*:|TotalNumVgprs|Occupancy|ScratchSize" /tmp/t/${1%.hip}.s | grep -v "^\s*;.*@" | paste - - - - | sed 's/_ZN12_GLOBAL__N_1[0-9]*//; s/; @_ZN[^ \t]*//' | cut -c1-150
