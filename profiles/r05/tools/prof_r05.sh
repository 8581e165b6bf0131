# Round-5 profile collection, as run on the GPU box (results -> gpurun_out/prof_r05/, copied into profiles/r05/).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r05
mkdir -p $O
# headline, one launch chain (every launch has the full 256-workgroup grid)
WMZ_CLIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_head1 -- python3 bench.py --steps 200 --warmup 20 --no-cone --train-steps 0 --no-cpu-baseline > $O/headline_1chain_bench_under_rocprof.json 2> $O/headline1_bench.err
cp $(ls /tmp/p_head1/*/*_kernel_stats.csv | head -1) $O/headline_1chain_kernel_stats.csv
# the training step at config 4 (8 eager steps)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train -- python3 tools/prof_train.py 8 > $O/train.log 2>&1
cp $(ls /tmp/p_train/*/*_kernel_stats.csv | head -1) $O/train_kernel_stats.csv
# the conv path: frame encoder (6 eager calls of 256 frames), VQ-AE training step (6 eager steps; 6 graphed steps)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_enc -- python3 tools/prof_encode.py 6 > $O/encode.log 2>&1
cp $(ls /tmp/p_enc/*/*_kernel_stats.csv | head -1) $O/frame_encoder_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vqae -- python3 tools/prof_vqae_train.py 6 > $O/vqae.log 2>&1
cp $(ls /tmp/p_vqae/*/*_kernel_stats.csv | head -1) $O/vqae_train_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vqaeg -- python3 tools/prof_vqae_graph.py 6 > $O/vqae_graph.log 2>&1
cp $(ls /tmp/p_vqaeg/*/*_kernel_stats.csv | head -1) $O/vqae_train_graph_kernel_stats.csv
# config 5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_sparse -- python3 tools/prof_sparse.py > $O/sparse.log 2>&1 || true
cp $(ls /tmp/p_sparse/*/*_kernel_stats.csv | head -1) $O/sparse_kernel_stats.csv || true
# PMC passes over the conv kernels (separate passes; --kernel-trace only beside --pmc): fabric bytes and issue counters
python3 tools/pmc_quick.py FETCH_SIZE conv -- python3 tools/prof_encode.py 3 > $O/conv_fetch.log 2>&1
python3 tools/pmc_quick.py WRITE_SIZE conv -- python3 tools/prof_encode.py 3 > $O/conv_write.log 2>&1
python3 tools/pmc_quick.py SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE conv -- python3 tools/prof_encode.py 3 > $O/conv_issue.log 2>&1
python3 tools/pmc_quick.py FETCH_SIZE convw -- python3 tools/prof_vqae_train.py 3 > $O/convw_fetch.log 2>&1
python3 tools/pmc_quick.py WRITE_SIZE convw -- python3 tools/prof_vqae_train.py 3 > $O/convw_write.log 2>&1
python3 tools/pmc_quick.py SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE convw -- python3 tools/prof_vqae_train.py 3 > $O/convw_issue.log 2>&1
python3 tools/time_conv.py 256 > $O/time_conv_256frames.txt 2>&1
python3 tools/pmc_traffic.py $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
python3 tools/pmc_issue.py $O/pmc_issue.json > $O/pmc_issue.log 2>&1
echo PROF_DONE
