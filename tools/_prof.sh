set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_70 -o l70 --output-format csv -- python3 $R/bench.py --model llama3-70b --no-sweep --no-cpu-baseline --steps 6 --warmup 2 > $R/gpurun_out/prof_70.json 2> $R/gpurun_out/prof_70.err
cd $R
python tools/decode_step_profile.py gpurun_out/prof_70/l70_kernel_trace.csv
