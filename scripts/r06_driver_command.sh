# The driver's own command, as the round-end bench runs it: stdout = the compact contract line (what BENCH_rNN.json keeps), wall clock beside it.
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r06_driver
T0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver/stdout.json 2> gpurun_out/r06_driver/stderr.log
T1=$(date +%s.%N)
python3 -c "print(\"wall_seconds\", round($T1 - $T0, 1))" | tee gpurun_out/r06_driver/wall.txt
cp gpurun_out/bench_full.json gpurun_out/r06_driver/bench_full.json
tail -c 1400 gpurun_out/r06_driver/stdout.json; echo; wc -c gpurun_out/r06_driver/stdout.json
