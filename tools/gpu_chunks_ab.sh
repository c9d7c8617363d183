cd ${GRAFT_REPO_ROOT:-/root/repo}
for ch in 64 128 250; do for sh in 0 8; do
 echo -n "chunks=$ch shard_of=$sh: "; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunks $ch --emulate-shard-of $sh | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done; done
