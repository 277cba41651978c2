for p in -1 0 1; do
echo "== prio $p (torchrun, process group up)"
MDCONV_FORK_PRIO=$p python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$((p+3)) tools/dist_overhead.py 2>/dev/null | grep " ms" | head -3
echo "== prio $p (plain bench)"
MDCONV_FORK_PRIO=$p python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
done
