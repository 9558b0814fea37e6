set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/r01f_bench.json 2> $R/gpurun_out/r01f_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f -o r01f --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 16 --warmup 4 > $R/gpurun_out/r01f_bench_under_rocprof.json 2> $R/gpurun_out/prof_f.err
export FERRUM_HIP_NO_GRAPH=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_f -o f --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $R/gpurun_out/pmc_fetch_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_f -o w --output-format csv -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $R/gpurun_out/pmc_write_f.err
cd $R
python tools/decode_step_profile.py gpurun_out/prof_f/r01f_kernel_trace.csv > gpurun_out/r01f_decode_step.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch_f > gpurun_out/r01f_pmc_fetch.txt
python tools/pmc_summary.py gpurun_out/pmc_write_f > gpurun_out/r01f_pmc_write.txt
head -12 gpurun_out/r01f_decode_step.txt
grep -A1 "w4_gemm_kernel<1, false, 2, 1> grid=221184" gpurun_out/r01f_pmc_fetch.txt gpurun_out/r01f_pmc_write.txt
