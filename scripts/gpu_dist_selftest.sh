mkdir -p gpurun_out
timeout -k 10 600 python bench.py --cells 128 --steps 50 --dist-selftest > gpurun_out/dist_selftest.json 2> gpurun_out/dist_selftest.err
rc=$?; cat gpurun_out/dist_selftest.json; tail -5 gpurun_out/dist_selftest.err
