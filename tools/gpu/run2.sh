#!/bin/bash
# v4 ablations + SQ counters of the conv micro-benchmark
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout -k 5 120 tools/bin/v4_diag 64 256 40 3 > $O/diag_256_40.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 512 40 3 > $O/diag_512_40.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 512 20 3 > $O/diag_512_20.log 2>&1
cd /tmp && export TMPDIR=/tmp
export CB_ONLY="256->256 @"
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o m -- $GRAFT_REPO_ROOT/tools/bin/conv_bench 3 64 > $O/pmc_m.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_mfma.py $(find $O/pmc_m -name "*counter_collection.csv" | head -1) $O/conv_mfma_util.csv > $O/conv_mfma_util.log 2>&1
rm -rf $O/pmc_m
echo done
