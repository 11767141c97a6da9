mkdir -p gpurun_out
timeout -k 10 600 python scripts/tune_spmv.py 256 0x0,0x1000,0x2000 > gpurun_out/tune256c.txt 2>&1
cat gpurun_out/tune256c.txt | tail -5
