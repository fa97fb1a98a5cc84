#!/bin/bash
# tools/mkvariant.sh NAME [-DFLAG ...]: libtrl_hip_NAME.so with k_ppo.hip (or $SRC) recompiled under the flags, every other
# object from /tmp/objs (built once per container: see NOTES_r06).  Development aid for within-session A/Bs.
set -e
NAME=$1; shift
SRC=${SRC:-k_ppo}
R=/root/repo
mkdir -p /tmp/objs/$NAME
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $R/torchrl_amd/csrc/$SRC.hip -o /tmp/objs/$NAME/$SRC.o
OBJS=$(ls /tmp/objs/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/torchrl_amd/lib/libtrl_hip_$NAME.so /tmp/objs/$NAME/$SRC.o $OBJS -ldl
echo built libtrl_hip_$NAME.so
