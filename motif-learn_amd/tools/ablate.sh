#!/bin/bash
# Times the batch kernel and its timing-only ablation builds (make -C motif-learn_amd/csrc ablate) with
# bench.py's HIP-event kernel timer.  Usage: motif-learn_amd/tools/ablate.sh  (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
L=$R/motif-learn_amd/mtflearn_amd/lib
for v in "" _ablate1 _ablate2 _ablate3 _v_base _v_ntload _v_ntstore ""; do
  MTFLEARN_AMD_LIB=$L/libzernike_hip$v.so python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense 2>/dev/null \
    | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant=%-10s kernel_ms=%.3f  %.0f GB/s algorithmic' % ('${v:-full}', d['roofline']['kernel_ms'], d['roofline']['achieved']))"
done
