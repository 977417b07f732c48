set -e
bash tools/collect_profiles_r03.sh 2>&1 | tail -30
python bench.py > gpurun_out/prof_r03/r03_bench.json 2> gpurun_out/prof_r03/r03_bench.err
bash tools/cycle_trace.sh > gpurun_out/prof_r03/cycle_trace.out 2>&1 || true
cp gpurun_out/cyc/r03_cycle_timeline.txt gpurun_out/cyc/r03_cycle_timeline_rks.txt gpurun_out/prof_r03/ 2>/dev/null || true
python tools/run_config.py ibuprofen def2-TZVP B3LYP --opt=50 > gpurun_out/prof_r03/r03_config5_ibuprofen_b3lyp_def2tzvp_opt.log 2>&1 || true
python tools/step_profile.py > gpurun_out/prof_r03/r03_step_profile.log 2>&1 || true
echo done
