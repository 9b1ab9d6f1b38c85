#!/bin/bash
# Register report of the 8-phase GEMM after an edit, in ~40 s instead of a 3-minute library build: compiles gemm.hip with -DALPRO_ISA_QUICK
# (two epilogue forms per dtype) into a scratch object and prints vgpr / sgpr / spill counts (tools/isa_report.py reads alpro_amd/lib/obj,
# so the scratch object is reported from its own directory).
set -e
D=/tmp/isa_quick; mkdir -p $D/alpro_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DALPRO_ISA_QUICK -c alpro_amd/csrc/gemm.hip -o $D/alpro_amd/lib/obj/gemm.o > $D/cc.log 2>&1 || { tail -30 $D/cc.log; exit 1; }
mkdir -p $D/tools && cp tools/isa_report.py $D/tools/
python $D/tools/isa_report.py gemm ${1:-nt256q}
