set -e
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2g/ktfull -- python3 bench.py > gpurun_out/r2g/bench_full_under_rocprof.json 2> gpurun_out/r2g/ktfull.err
find gpurun_out/r2g/ktfull -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2g/kernel_stats_full.csv \;
rm -rf gpurun_out/r2g/ktfull
timeout -k 10 600 python3 bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
timeout -k 10 600 python3 bench.py --gpus 1 --spawn > gpurun_out/r2g/bench_spawn.json 2> gpurun_out/r2g/bench_spawn.err || true
tail -c 300 gpurun_out/r2g/bench.json
