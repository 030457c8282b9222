#!/bin/bash
# the table of DESIGN.md section 8: step time with C stand-in communication workgroups, launches sized for 256 CUs and for 256 - C
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/comm_standin.jsonl; : > $OUT
python tools/comm_standin.py --c 0 >> $OUT 2>gpurun_out/r4/comm_standin.err
for c in 8 16 32; do
  python tools/comm_standin.py --c $c >> $OUT 2>>gpurun_out/r4/comm_standin.err
  GIPVIT_CU_BUDGET=$((256 - c)) python tools/comm_standin.py --c $c >> $OUT 2>>gpurun_out/r4/comm_standin.err
done
GIPVIT_CU_BUDGET=248 python tools/comm_standin.py --c 0 >> $OUT 2>>gpurun_out/r4/comm_standin.err
cat $OUT
