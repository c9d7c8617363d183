R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for b in 64 128 256; do make -s -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_b$b.so EXTRA="-DRTW_ITEM_BATCH=${b}u" 2>&1 | grep error; done
for b in 64 128 256; do for ch in 64; do for sh in 0 8; do
 echo -n "batch=$b chunks=$ch shard_of=$sh: "; RTW_HIP_LIB=/tmp/librtw_b$b.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunks $ch --emulate-shard-of $sh | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'])"
done; done; done
