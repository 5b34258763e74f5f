# Re-takes everything under profiles/r05_* on a GPU box (run through gpurun from the repository root, in two calls):
#   gpurun --timeout 1150 -- 'bash tools/bench_scripts/regen_profiles_r05.sh pmc'
#   gpurun --timeout 1150 -- 'bash tools/bench_scripts/regen_profiles_r05.sh trace'
# One rocprofv3 --pmc run per counter group, never combined with tracing; every step under its own timeout, joined with &&.
set -e
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
if [ "$1" = "pmc" ]; then
timeout -k 10 500 bash profiles/pmc_pass.sh $O/pmc && python3 profiles/pmc_summarise.py $O/pmc $O/pmc_summary.json > /dev/null && rm -rf $O/pmc/p? && echo "pmc headline done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_crowded python3 tools/bench_scripts/crowded_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_crowded $O/pmc_crowded_summary.json ordered_ > /dev/null && rm -rf $O/pmc_crowded/p? && echo "pmc crowded done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_gamma python3 tools/bench_scripts/gamma_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_gamma $O/pmc_gamma_summary.json ordered_ > /dev/null && rm -rf $O/pmc_gamma/p? && echo "pmc gamma done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_khist_noise python3 tools/bench_scripts/prof_kmeans_hist.py 32 noise && python3 profiles/pmc_summarise.py $O/pmc_khist_noise $O/pmc_khist_noise_summary.json hist_ > /dev/null && rm -rf $O/pmc_khist_noise/p? && echo "pmc kmeans noise done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_khist_smooth python3 tools/bench_scripts/prof_kmeans_hist.py 32 smooth && python3 profiles/pmc_summarise.py $O/pmc_khist_smooth $O/pmc_khist_smooth_summary.json hist_ > /dev/null && rm -rf $O/pmc_khist_smooth/p? && echo "pmc kmeans smooth done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_kpix python3 tools/bench_scripts/prof_kmeans.py 32 && python3 profiles/pmc_summarise.py $O/pmc_kpix $O/pmc_kpix_summary.json kmeans_ > /dev/null && rm -rf $O/pmc_kpix/p? && echo "pmc kmeans over pixels done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_ed python3 tools/bench_scripts/ed_prof.py 16 256 && python3 profiles/pmc_summarise.py $O/pmc_ed $O/pmc_ed_summary.json ed_wavefront > /dev/null && rm -rf $O/pmc_ed/p? && echo "pmc ed done" && \
timeout -k 10 300 bash profiles/pmc_pass_cmd4.sh $O/pmc_c5 python3 tools/bench_scripts/c5_prof.py 4 && python3 profiles/pmc_summarise.py $O/pmc_c5 $O/pmc_c5_summary.json ordered_ > /dev/null && rm -rf $O/pmc_c5/p? && echo "pmc c5 done"
else
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-extra --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err && \
python3 profiles/trace_headline.py $O/kt $O/trace_headline.csv && \
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; && find $O/kt -name "*kernel_trace.csv" -delete && \
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktfull -- python3 bench.py > $O/bench_full_under_rocprof.json 2> $O/ktfull.err && \
find $O/ktfull -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_full.csv \; && rm -rf $O/ktfull && \
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err && \
find $O -name "*.csv" -size +2M -delete && tail -c 400 $O/bench.json
fi
