mkdir -p gpurun_out
for n in 128 256; do
timeout -k 10 900 python scripts/mg_probe.py $n > gpurun_out/mg_probe$n.txt 2>&1
tail -5 gpurun_out/mg_probe$n.txt
done
