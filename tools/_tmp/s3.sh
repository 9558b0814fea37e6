set -e -o pipefail
run() { echo "== $ENVS $*"; timeout -k 10 400 python3 bench.py --no-sweep --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
ENVS="base"; run
export FERRUM_HIP_ROUTE_PARTS=2; ENVS="Q=2"; run
export FERRUM_HIP_ROUTE_PARTS=8; ENVS="Q=8"; run
unset FERRUM_HIP_ROUTE_PARTS
export FERRUM_HIP_ATTN_NARROW=1; ENVS="attn NW=4"; run
unset FERRUM_HIP_ATTN_NARROW
export FERRUM_HIP_ATTN_SPLITS=2; ENVS="attn splits=2"; run
unset FERRUM_HIP_ATTN_SPLITS
export FERRUM_HIP_MOE_EM_PAIRS=0; ENVS="block-major"; run
echo ALL-OK
