#!/bin/bash
# Two bench.py ranks on ONE GPU over gloo: exercises the N > 1 path of bench.py (size exchange, narrow count
# buffers, double-buffered asynchronous gather, max-over-ranks timing) with the real kernels when no multi-GPU
# node is at hand.  RCCL itself is not involved.  usage: tools/dryrun_two_ranks.sh [bench args, default cfg2]
ARGS=${*:---workload cfg2 --steps 4}
GDX_BENCH_ONE_GPU=1 GDX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 $ARGS
