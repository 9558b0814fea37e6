#!/bin/bash
# Round evidence in one gpurun call: the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# the two PMC passes for roofline.traffic (FETCH_SIZE / WRITE_SIZE, separate runs, eager launches), and the extra models.
# Usage (on the GPU box): bash tools/collect_round_profiles.sh r02x
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-tp-scaling > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
echo "trace done"
python3 tools/decode_step_db.py $OUT/trace/t_results.db > $OUT/decode_step.txt || exit 1
FERRUM_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 4 --warmup 2 --layers 14 --no-sweep --no-cpu-baseline --no-tp-scaling > /dev/null 2> $OUT/pmc_fetch.err || exit 1
FERRUM_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py --steps 4 --warmup 2 --layers 14 --no-sweep --no-cpu-baseline --no-tp-scaling > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 tools/pmc_summary.py $OUT/pmc_fetch w4_gemm_moe_em > $OUT/pmc_fetch_size.txt
python3 tools/pmc_summary.py $OUT/pmc_write w4_gemm_moe_em > $OUT/pmc_write_size.txt
echo "pmc done"
for m in llama31-8b gemma3-27b llama3-70b; do
  python3 bench.py --model $m --no-sweep --no-cpu-baseline --steps 32 --warmup 4 > $OUT/bench_$m.json 2> $OUT/bench_$m.err || exit 1
done
rm -rf $OUT/pmc_fetch $OUT/pmc_write
echo ALL-OK
