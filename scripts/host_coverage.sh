#!/bin/bash
# Which lines / branches of the product's HOST code (zl_engine.cpp, zl_libzl.cpp and the headers they include: zl_host.h, zl_sched.h,
# zl_handoff.h) do the GPU tests execute?  The two files are rebuilt with g++ --coverage, linked with the shipped kernels object into
# a scratch library the tests load through ZLHIP_LIBRARY, and gcov -b writes its listing to gpurun_out/host_cov/.  Run on a GPU box:
#   gpurun -- 'mkdir -p gpurun_out/host_cov && bash scripts/host_coverage.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
W=/tmp/zl_hcov; rm -rf $W; mkdir -p $W $R/gpurun_out/host_cov
cd $W || exit 1
for f in zl_engine zl_libzl; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --coverage -w -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I $R/include -I $R/libzl_amd/csrc \
      -c $R/libzl_amd/csrc/$f.cpp -o $f.o || exit 1
done
K=${ZL_KERNELS_OBJ:-$R/libzl_amd/lib/obj/libzlhip/zl_kernels.o}; [ -f $K ] || K=$R/gpurun_in/zl_kernels.o   # (the object cache does not travel with gpurun: copy it to gpurun_in/ first)
[ -f $K ] || { echo "no kernels object ($K): run python -m libzl_amd.build first (the object cache does not travel with gpurun)"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $W/libzlhip_cov.so $K zl_engine.o zl_libzl.o -lgcov || exit 1
( cd $R && ZLHIP_LIBRARY=$W/libzlhip_cov.so timeout -k 10 900 python -m pytest tests -q -m "${TIER:-gpu}" 2>&1 | tail -2 )
gcov -b -o $W $R/libzl_amd/csrc/zl_engine.cpp $R/libzl_amd/csrc/zl_libzl.cpp 2>&1 | grep -A3 "File '$R" | grep -v "^--" | tee $R/gpurun_out/host_cov/summary.txt
cp zl_engine.cpp.gcov zl_libzl.cpp.gcov zl_host.h.gcov zl_sched.h.gcov zl_handoff.h.gcov $R/gpurun_out/host_cov/ 2>/dev/null
