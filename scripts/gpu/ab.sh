PPF_FORCE_GRADSYNC=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-300
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
