#!/bin/bash
# gemm_os_kernel with parts switched off (MMLREC_OS_LAB: 1 no MFMAs, 2 no cut, 4 no weight fragment reads, 8 no gradient
# fragment reads, 16 no DMA / waits, 32 no barrier; sums combine): where a k-step's time goes
cd $GRAFT_REPO_ROOT
# needs a lab build of the library: MMLREC_BUILD_OS_LAB=1 MMLREC_FORCE_BUILD=1 python3 -c 'import __graft_entry__ as g; g.build()' (then rebuild without)
for lab in 0 1 2 4 8 12 14 16 32 48 62 15; do
  echo "LAB $lab: $(MMLREC_OS_LAB=$lab python3 tools/lab/os_time.py 2>&1 | grep gemm_os | head -2 | tail -1 | sed 's/.*gemm_os_kernel//')"
done
