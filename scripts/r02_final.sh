#!/bin/bash
# End-of-round evidence: the whole GPU suite, the profile set (scripts/r02_profiles.sh), the decode A/B at 1 / 8 / 16 sequences.
OUT=$(pwd)/gpurun_out/r02_final; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash scripts/r02_profiles.sh r02_final/prof 2>&1 | tail -40
timeout 600 python scripts/decode_ab.py $OUT/decode_ab.json 1 8 16 > $OUT/decode_ab.log 2>&1; grep -E "^==" $OUT/decode_ab.log | cut -c1-200
