#!/bin/bash
# round 6, end of round: the full GPU suite + smoke() on the tree that is committed, then the evidence set
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash scripts/r06_profiles.sh r06_final
