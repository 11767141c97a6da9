mkdir -p gpurun_out
MXG_SHARE_GPU=1 NCCL_DEBUG=WARN timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --cells 64 --steps 20 --warmup 2 > gpurun_out/dist_2rank_1gpu.json 2> gpurun_out/dist_2rank_1gpu.err
echo "2-rank rc=$?"; cat gpurun_out/dist_2rank_1gpu.json; grep -i "error\|duplicate\|invalid\|WARN" gpurun_out/dist_2rank_1gpu.err | head -8
