#!/bin/bash
# round 2, GPU call 1: the new full-depth parity + free-running greedy tests, then the default bench line (metric config)
OUT=$(pwd)/gpurun_out/r02_run1; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_fulldepth_parity_gpu.py tests/test_e2e_gpu.py::test_free_running_greedy_ids_vs_oracle -m gpu -q --timeout 800 > $OUT/pytest_new.log 2>&1; tail -30 $OUT/pytest_new.log
cp gpurun_out/fulldepth_metrics.json $OUT/ 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json; tail -5 $OUT/bench.err
