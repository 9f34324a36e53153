set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_full.py tests/test_gpu_fullsize.py -x -q --timeout 600 --timeout-method thread -k "beam" > gpurun_out/r2_t19.log 2>&1
timeout -k 10 600 python bench.py --config largev3_beam5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_beam_new.json 2> gpurun_out/r2_bench_x.err
