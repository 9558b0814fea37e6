set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FERRUM_HIP_NO_GRAPH=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_l -o p --output-format csv -- python3 $R/bench.py --model llama31-8b --no-sweep --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc_l.err
cd $R
python tools/pmc_summary.py gpurun_out/pmc_l ldsa
python tools/pmc_summary.py gpurun_out/pmc_l "w4_gemm_kernel<1"
