set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c1 -o c1 --output-format csv -- python3 $R/bench.py --concurrency 1 --no-sweep --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/prof_c1.json 2> $R/gpurun_out/prof_c1.err
cd $R
python tools/decode_step_profile.py gpurun_out/prof_c1/c1_kernel_trace.csv
