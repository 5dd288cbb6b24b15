# Run ON THE GPU BOX: MFMA / LDS utilisation counters of gemm_nt_kernel on the step's hot shapes (tools/pmc_gemm.py),
# one rocprofv3 --pmc pass per counter group (kernel-trace only, as gpurun requires).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gemm_pmc
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_gemm.py > $OUT/p$i.log 2>&1 || echo "group '$grp' failed" >> $OUT/failed.txt
done
ls $OUT
