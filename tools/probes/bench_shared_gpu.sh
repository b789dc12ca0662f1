#!/bin/bash
# bench.py's N > 1 flow exactly as the driver launches it, with all ranks on the box's ONE GPU over the peer-to-peer transport
# (RCCL refuses ranks that share a device): the N = 2 and N = 8 lines with their multi_gpu / parity objects.  Not a scaling
# measurement — the ranks share 256 CUs — but the whole multi-rank code path on the HIP library.  (gpurun, repository root)
OUT=$(pwd)/gpurun_out/shared
mkdir -p $OUT
for n in ${SHARED_N:-2 8}; do
  TNN_COMM=xgmi TNN_DEVICE=0 TNN_P2P_TIMEOUT_MS=20000 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python3 -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 128 --warmup 64 \
    > $OUT/benchA_dp${n}_shared_gpu.json 2> $OUT/benchA_dp${n}_shared_gpu.err
  echo "N=$n rc=$? $(tail -c 200 $OUT/benchA_dp${n}_shared_gpu.json | head -c 120)"
done
