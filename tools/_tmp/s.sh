set -e -o pipefail
run() { echo "== $*"; timeout -k 10 400 python3 bench.py --no-sweep --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('e2e_tok_s'))"; }
run --model llama31-8b --concurrency 48
export FERRUM_HIP_W4_TILE_MIN_M=33
echo "TILE_MIN_M=33"
run --model gemma3-27b --concurrency 48
run --model llama31-8b --concurrency 48
run --concurrency 48
