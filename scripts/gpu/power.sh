timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
BP=$!
for i in $(seq 1 30); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 1; kill -0 $BP 2>/dev/null || break; done
wait $BP; cut -c90-200 /tmp/b.json
