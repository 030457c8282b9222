mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/comm_standin_p6.jsonl; : > $OUT
for c in 8 16; do
  python tools/comm_standin.py --c $c --passes 6 >> $OUT 2>>gpurun_out/r4/comm_standin.err
  GIPVIT_CU_BUDGET=$((256 - c)) python tools/comm_standin.py --c $c --passes 6 >> $OUT 2>>gpurun_out/r4/comm_standin.err
done
python tools/comm_standin.py --c 8 --passes 12 >> $OUT 2>>gpurun_out/r4/comm_standin.err
GIPVIT_CU_BUDGET=248 python tools/comm_standin.py --c 8 --passes 12 >> $OUT 2>>gpurun_out/r4/comm_standin.err
cat $OUT
