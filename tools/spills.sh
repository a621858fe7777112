#!/bin/bash
# tools/spills.sh <kernel pattern>: compile rg_mpc.hip to ISA and list the kernel's spill stores with their producers
cd $(dirname $0)/../robot_gym_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -S --cuda-device-only -o /tmp/rg_mpc.s rg_mpc.hip 2>/dev/null
cd ../.. && python3 tools/isa_scratch.py /tmp/rg_mpc.s "$1" --defs
