set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/headline -- python3 bench.py --steps 200 --warmup 20 --no-cone --train-steps 0 --no-cpu-baseline > gpurun_out/prof_r03/headline_bench.json 2> gpurun_out/prof_r03/headline_bench.err
python3 tools/trace_by_grid.py gpurun_out/prof_r03/headline gpurun_out/prof_r03/headline_by_grid.csv
# the same with ONE launch chain (config.clip_streams = 1): every launch of the trace has the full 256-workgroup grid, so the
# plain --stats rows are directly the headline shape
WMZ_CLIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/headline1 -- python3 bench.py --steps 200 --warmup 20 --no-cone --train-steps 0 --no-cpu-baseline > gpurun_out/prof_r03/headline1_bench.json 2> gpurun_out/prof_r03/headline1_bench.err
cp $(ls gpurun_out/prof_r03/headline1/*/*_kernel_stats.csv | head -1) gpurun_out/prof_r03/headline_1chain_kernel_stats.csv
cp $(ls gpurun_out/prof_r03/headline/*/*_kernel_stats.csv | head -1) gpurun_out/prof_r03/headline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/train -- python3 tools/prof_train.py 8 > gpurun_out/prof_r03/train.log 2>&1
cp $(ls gpurun_out/prof_r03/train/*/*_kernel_stats.csv | head -1) gpurun_out/prof_r03/train_kernel_stats.csv
python3 tools/pmc_traffic.py gpurun_out/prof_r03/pmc_traffic.json > gpurun_out/prof_r03/pmc_traffic.log 2>&1
python3 tools/pmc_issue.py gpurun_out/prof_r03/pmc_issue.json > gpurun_out/prof_r03/pmc_issue.log 2>&1
echo PROF_DONE
