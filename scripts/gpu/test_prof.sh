timeout 800 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E |passed|failed|Error|error|^tests" | cut -c1-400
bash scripts/gpu/prof.sh
