set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p -o pf --output-format csv -- python3 $R/tools/prefill_only.py > $R/gpurun_out/prefill.log 2> $R/gpurun_out/prefill.err
