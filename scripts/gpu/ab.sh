timeout 600 python scripts/gpu/host_time.py 2>&1 | tail -45 | cut -c1-160
