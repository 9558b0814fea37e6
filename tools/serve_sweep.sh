#!/bin/bash
# Robustness sweep of bin/ferrum_hip_serve over configurations beyond the bench default (development aid).  Stops at the
# first failure; every run must return every KV block to the pool ("kv_free_blocks" == "kv_total_blocks").
set -e -o pipefail
S=./ferrum-infer-rs_amd/bin/ferrum_hip_serve
run() { echo "== $*" ; timeout -k 10 300 $S "$@" | tail -1 ; }
run --layers 8 --requests 64 --concurrency 64 --out-len 96 --out-len-jitter 32
run --layers 8 --requests 48 --concurrency 16 --prompt-len 1000 --out-len 40 --max-batched-tokens 512
run --layers 8 --requests 40 --concurrency 24 --prompt-len 300 --out-len 150 --out-len-jitter 100 --kv-blocks 420
run --layers 8 --requests 30 --concurrency 100 --prompt-len 64 --out-len 200 --out-len-jitter 50
run --layers 8 --requests 12 --concurrency 3 --prompt-len 3000 --out-len 20 --max-batched-tokens 2048
run --layers 8 --requests 64 --concurrency 48 --prompt-len 17 --out-len 33 --out-len-jitter 16 --max-batched-tokens 64
run --dense --layers 8 --requests 64 --concurrency 64 --out-len 96 --out-len-jitter 32
run --dense --layers 8 --requests 40 --concurrency 20 --prompt-len 700 --out-len 60 --out-len-jitter 30 --max-batched-tokens 1024
echo ALL-OK
