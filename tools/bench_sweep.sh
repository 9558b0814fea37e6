#!/bin/bash
# Fault triage / robustness: bench.py at batch sizes and models beyond the default (development aid).  Stops at the first failure.
set -e -o pipefail
run() { echo "== $*"; timeout -k 10 400 python3 bench.py --no-sweep --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('e2e_tok_s'))"; }
run --concurrency 17
run --concurrency 33 --steps 100
run --concurrency 48 --steps 128 --warmup 3
run --concurrency 2 --steps 200 --warmup 1
run --model llama31-8b --concurrency 64
run --model llama31-8b --concurrency 17 --steps 100
run --model gemma3-27b --concurrency 48
run --model llama3-70b --concurrency 5 --steps 40
echo ALL-OK
