mkdir -p gpurun_out
timeout -k 10 600 python bench.py --cells 128 --steps 50 --dist-selftest > gpurun_out/dist_selftest.json 2> gpurun_out/dist_selftest.err
rc=$?; cat gpurun_out/dist_selftest.json; tail -5 gpurun_out/dist_selftest.err
if [ $rc -ge 124 ]; then exit $rc; fi
# two ranks on the one GPU of this box: expected to be refused by RCCL (duplicate device); recorded for DESIGN.md
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --cells 64 --steps 20 --warmup 2 > gpurun_out/dist_2rank_1gpu.json 2> gpurun_out/dist_2rank_1gpu.err
echo "2-rank rc=$?"; cat gpurun_out/dist_2rank_1gpu.json; grep -i "error\|duplicate\|invalid" gpurun_out/dist_2rank_1gpu.err | head -5
