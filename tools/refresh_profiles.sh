# manual helper: copy the latest tools/gpu/run_round.sh outputs from gpurun_out/rNN into profiles/rNN.  usage: refresh_profiles.sh r02
R=${1:-r02}; O=gpurun_out/$R; P=profiles/$R
newest() { ls -t $1 | head -1; }
cp $O/bench.json $P/bench_$R.json
cp $O/bench_prof.json $P/bench_under_rocprof.json
cp "$(newest "$O/prof/*/*kernel_stats.csv")" $P/bench_kernel_stats.csv
cp "$(newest "$O/prof2048/*/*kernel_stats.csv")" $P/bench_2048_kernel_stats.csv
cp "$(newest "$O/pmc_f/*/*counter_collection.csv")" $P/pmc_fetch_size.csv
cp "$(newest "$O/pmc_w/*/*counter_collection.csv")" $P/pmc_write_size.csv
cp $O/bench_rank.json $P/bench_launcher_world1.json
cp $O/bench_2048.json $P/bench_2048_$R.json
python tools/pmc_traffic.py $P/pmc_fetch_size.csv $P/pmc_write_size.csv $P/pmc_traffic.json 32
