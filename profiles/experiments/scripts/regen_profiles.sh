set -e
mkdir -p gpurun_out/r2f
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/gpu_tests.log 2>&1; tail -2 gpurun_out/r2f/gpu_tests.log
timeout -k 10 600 bash profiles/pmc_pass.sh gpurun_out/r2f/pmc
python3 profiles/pmc_summarise.py gpurun_out/r2f/pmc gpurun_out/r2f/pmc_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/kt -- python3 bench.py --no-extra --no-cpu-baseline > gpurun_out/r2f/bench_under_rocprof.json 2> gpurun_out/r2f/kt.err
python3 profiles/trace_headline.py gpurun_out/r2f/kt gpurun_out/r2f/trace_headline.csv
find gpurun_out/r2f/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2f/kernel_stats.csv \;
find gpurun_out/r2f/kt -name "*kernel_trace.csv" -delete
find gpurun_out/r2f/pmc -name "*.csv" -size +2M -delete
cat gpurun_out/r2f/bench_under_rocprof.json | tail -1 | cut -c1-600
