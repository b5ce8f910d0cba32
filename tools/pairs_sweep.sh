#!/bin/bash
# GPU parity suite, then bench.py --pairs over the tile configurations (two runs each).
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_pairs.log 2>&1 || { tail -30 gpurun_out/pytest_pairs.log; exit 1; }
tail -3 gpurun_out/pytest_pairs.log
for cfg in 0 1 2 3 4 5; do
  for rep in 1 2; do
    out=$(timeout -k 10 200 python bench.py --pairs --log2-keys 27 --tile-config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "cfg $cfg $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"], d["config"].get("tile_keys"))')"
  done
done
