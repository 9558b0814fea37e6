set -e -o pipefail
run() { echo "== $*"; timeout -k 10 400 python3 bench.py --no-sweep --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('e2e_tok_s'), d.get('prefill_ms_all_prompts'))"; }
run --prompt-len 1000 --prefill-chunk 2048 --steps 40
run --prompt-len 100 --prefill-chunk 512 --concurrency 40 --steps 30
run --prompt-len 3000 --concurrency 8 --steps 300 --warmup 2
run --prompt-len 17 --concurrency 32 --steps 500 --warmup 1
run --model llama31-8b --prompt-len 2000 --concurrency 16 --prefill-chunk 4096 --steps 200
run --model gemma3-27b --prompt-len 1500 --concurrency 8 --steps 100
echo ALL-OK
