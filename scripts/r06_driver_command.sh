export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r06_driver
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver/stdout.json 2> gpurun_out/r06_driver/stderr.log
cp gpurun_out/bench_full.json gpurun_out/r06_driver/bench_full.json
tail -c 2000 gpurun_out/r06_driver/stdout.json; grep "Elapsed" gpurun_out/r06_driver/stderr.log; wc -c gpurun_out/r06_driver/stdout.json
