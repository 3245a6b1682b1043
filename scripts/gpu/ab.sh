for t in 288 400 540 288 400 540; do
echo -n "target $t: "; PPF_SPLITK_TARGET=$t timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
done
