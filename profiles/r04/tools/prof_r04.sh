# Round-4 profile collection, as run on the GPU box (results -> gpurun_out/prof_r04/, copied into profiles/r04/).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04
mkdir -p $O
# headline only, one launch chain: every launch of the trace has the full 256-workgroup grid (the rows bench.py's per-kernel
# fractions can be recomputed from with no filtering), then the product's two clip chains
WMZ_CLIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline1 -- python3 bench.py --steps 200 --warmup 20 --no-cone --train-steps 0 --no-cpu-baseline > $O/headline_1chain_bench_under_rocprof.json 2> $O/headline1_bench.err
cp $(ls $O/headline1/*/*_kernel_stats.csv | head -1) $O/headline_1chain_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python3 bench.py --steps 200 --warmup 20 --no-cone --train-steps 0 --no-cpu-baseline > $O/headline_bench_under_rocprof.json 2> $O/headline_bench.err
cp $(ls $O/headline/*/*_kernel_stats.csv | head -1) $O/headline_kernel_stats.csv
python3 tools/trace_by_grid.py $O/headline $O/headline_by_grid.csv
# the training step at config 4 (8 eager steps), the published dim-384 model, the reference geometry is part of bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 tools/prof_train.py 8 > $O/train.log 2>&1
cp $(ls $O/train/*/*_kernel_stats.csv | head -1) $O/train_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train384 -- python3 tools/prof_train.py 4 384 512 20 3 1 1 > $O/train384.log 2>&1
cp $(ls $O/train384/*/*_kernel_stats.csv | head -1) $O/train_dim384_kernel_stats.csv
python3 tools/pmc_traffic.py $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
python3 tools/pmc_issue.py $O/pmc_issue.json > $O/pmc_issue.log 2>&1
echo PROF_DONE
