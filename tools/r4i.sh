mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/t4.log 2>&1; echo "tests rc $?"; tail -n 3 gpurun_out/r4/t4.log
python bench.py --arch vit_base --batch 64 --micro 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4/b4_vitb.json 2> gpurun_out/r4/b4_vitb.err; echo "vitb rc $?"
GIPVIT_CLS_ONLY_LAST=0 python bench.py --arch vit_base --batch 64 --micro 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4/b4_vitb_full.json 2> gpurun_out/r4/b4_vitb_full.err
python bench.py --config c2 --no-cpu-baseline > gpurun_out/r4/b4_c2.json 2> gpurun_out/r4/b4_c2.err
python -c "
import json
for f in ('b4_vitb','b4_vitb_full','b4_c2'):
    d=json.load(open('gpurun_out/r4/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['final_loss'], d['mfma_frac_whole_step'])"
