# bench.py's multi-rank code (users-DP through Recoder.train, union item count, max-over-ranks
# timing, the JSON line) with 2 processes on ONE GPU and gloo collectives: a code-path test for
# boxes with a single GPU -- the number it prints is meaningless and marked INVALID.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
  bench.py --gpus 2 --steps 20 --warmup 5 --one-gpu-gloo "$@"
