# Round-4 profile refresh after the wgrad3 rework (LDS-DMA ring) and the chain backward kernels: the training traces only (the
# headline kernels and their PMC passes are unchanged since tools/prof_r04.sh).  Results -> gpurun_out/prof_r04b/, copied into profiles/r04/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04b
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train -- python3 tools/prof_train.py 8 > $O/train.log 2>&1
cp $(ls /tmp/p_train/*/*_kernel_stats.csv | head -1) $O/train_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train384 -- python3 tools/prof_train.py 4 384 512 20 3 1 1 > $O/train384.log 2>&1
cp $(ls /tmp/p_train384/*/*_kernel_stats.csv | head -1) $O/train_dim384_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train96 -- python3 tools/prof_train.py 4 96 256 12 3 1 1 > $O/train96.log 2>&1
cp $(ls /tmp/p_train96/*/*_kernel_stats.csv | head -1) $O/train_dim96_kernel_stats.csv
python3 tools/pmc_quick.py FETCH_SIZE wgrad3 -- python3 tools/time_wgrad.py > $O/wgrad3_fetch.log 2>&1
python3 tools/pmc_quick.py WRITE_SIZE wgrad3 -- python3 tools/time_wgrad.py > $O/wgrad3_write.log 2>&1
python3 tools/pmc_quick.py SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE wgrad3 -- python3 tools/time_wgrad.py > $O/wgrad3_issue.log 2>&1
python3 tools/time_wgrad.py > $O/wgrad3_times.log 2>&1
echo PROF_DONE
