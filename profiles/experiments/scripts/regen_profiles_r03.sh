# Re-takes everything under profiles/r03_* on a GPU box (run through gpurun from the repository root):
#   gpurun --timeout 1150 -- 'bash tools/bench_scripts/regen_profiles_r03.sh'
# One rocprofv3 --pmc run per counter group, never combined with tracing; every step under its own timeout, joined with &&.
set -e
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 bash profiles/pmc_pass.sh $O/pmc && python3 profiles/pmc_summarise.py $O/pmc $O/pmc_summary.json > /dev/null && echo "pmc headline done" && \
timeout -k 10 400 bash profiles/pmc_pass_cmd.sh $O/pmc_crowded python3 tools/bench_scripts/crowded_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_crowded $O/pmc_crowded_summary.json ordered_ > /dev/null && echo "pmc crowded done" && \
timeout -k 10 400 bash profiles/pmc_pass_cmd.sh $O/pmc_gamma python3 tools/bench_scripts/gamma_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_gamma $O/pmc_gamma_summary.json ordered_ > /dev/null && echo "pmc gamma done" && \
( export DITHER_PIE_EXPERIMENTS=1 DP_NO_COMPACT_KERNEL=1; timeout -k 10 400 bash profiles/pmc_pass_cmd.sh $O/pmc_crowded_before python3 tools/bench_scripts/crowded_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_crowded_before $O/pmc_crowded_before_summary.json ordered_ > /dev/null && timeout -k 10 400 bash profiles/pmc_pass_cmd.sh $O/pmc_gamma_before python3 tools/bench_scripts/gamma_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_gamma_before $O/pmc_gamma_before_summary.json ordered_ > /dev/null ) && echo "pmc before done" && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-extra --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err && \
python3 profiles/trace_headline.py $O/kt $O/trace_headline.csv && \
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; && find $O/kt -name "*kernel_trace.csv" -delete && \
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktfull -- python3 bench.py > $O/bench_full_under_rocprof.json 2> $O/ktfull.err && \
find $O/ktfull -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_full.csv \; && rm -rf $O/ktfull && \
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err && \
find $O -name "*.csv" -size +2M -delete && tail -c 400 $O/bench.json
