mkdir -p gpurun_out/r4 gpurun_out/prof
bash tools/profile_step.sh r04 > gpurun_out/r4/profile_step.log 2>&1; echo "profile rc $?"; tail -n 5 gpurun_out/r4/profile_step.log
python bench.py > gpurun_out/r4/b6.json 2> gpurun_out/r4/b6.err; echo "bench rc $?"
python tools/phase_times.py > gpurun_out/r4/phase6.log 2>&1; tail -n 12 gpurun_out/r4/phase6.log
