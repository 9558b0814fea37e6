set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_d -o r01d --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 16 --warmup 4 > $R/gpurun_out/prof_d_bench.json 2> $R/gpurun_out/prof_d.err
export FERRUM_HIP_NO_GRAPH=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
python tools/decode_step_profile.py gpurun_out/prof_d/r01d_kernel_trace.csv > gpurun_out/r01d_decode_step.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch > gpurun_out/r01d_pmc_fetch.txt
python tools/pmc_summary.py gpurun_out/pmc_write > gpurun_out/r01d_pmc_write.txt
cat gpurun_out/r01d_decode_step.txt
grep -A2 "w4_gemm_kernel<1, false, 2>" gpurun_out/r01d_pmc_fetch.txt | head; grep -A2 "w4_gemm_kernel<1, false, 2>" gpurun_out/r01d_pmc_write.txt | head
