#!/bin/bash
# build_variant.sh NAME "-DPLM_X=.. ..."  ->  evcouplings_amd/libplm_NAME.so (kernel A/B experiments via PLM_HIP_LIB)
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/evcouplings_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $2"
/opt/rocm/bin/hipcc $F -c $SRC/plm_kernels.hip -o /tmp/pk_$1.o
/opt/rocm/bin/hipcc $F -x hip -c $SRC/plm_host.cpp -o /tmp/ph_$1.o
/opt/rocm/bin/hipcc $F -c $SRC/plm_meanfield.hip -o /tmp/pm_$1.o
/opt/rocm/bin/hipcc $F -x hip -c $SRC/plm_rccl.cpp -o /tmp/pr_$1.o
/opt/rocm/bin/hipcc $F -x hip -c $SRC/plm_io.cpp -o /tmp/pi_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/evcouplings_amd/libplm_$1.so /tmp/pk_$1.o /tmp/ph_$1.o /tmp/pm_$1.o /tmp/pr_$1.o /tmp/pi_$1.o -ldl
echo built libplm_$1.so
