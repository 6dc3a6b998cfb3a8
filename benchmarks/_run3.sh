mkdir -p gpurun_out/r3g; O=gpurun_out/r3g
python bench.py --mode train --steps 5 --warmup 2 > $O/train_n1.json 2> $O/train_n1.err; tail -c 700 $O/train_n1.json; echo
python bench.py --mode train --steps 5 --warmup 2 --force-dist-path > $O/train_ranks_path.json 2> $O/train_ranks_path.err; tail -c 600 $O/train_ranks_path.json; echo; tail -3 $O/train_ranks_path.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --mode train --steps 5 --warmup 2 --force-dist-path > $O/train_rccl1.json 2> $O/train_rccl1.err; tail -c 600 $O/train_rccl1.json; echo; tail -3 $O/train_rccl1.err
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('train_step'))
"
