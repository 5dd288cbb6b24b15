# Run ON THE GPU BOX (gpurun): counters for gemm256_kernel - the kernel that dominates the step - on the four big
# step shapes (tools/pmc_gemm256.py).  One rocprofv3 --pmc pass per counter group with --kernel-trace only (gpurun refuses --pmc
# together with the sys / hip / hsa trace domains); FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots, MI355X_MICROARCH.md).
# tools/summarise_gemm256_pmc.py reduces gpurun_out/prof_gemm256_pmc to profiles/r04_gemm256_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gemm256_pmc
rm -rf $OUT && mkdir -p $OUT
export VLA_PMC_META=$GRAFT_REPO_ROOT/$OUT/meta.json
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_gemm256.py > $OUT/p$i.log 2>&1 || echo "group '$grp' failed" >> $OUT/failed.txt
done
# an un-profiled timing of the same launches (kernel-trace only) for the durations
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/pmc_gemm256.py > $OUT/trace.log 2>&1
python3 tools/summarise_gemm256_pmc.py --stage box
ls $OUT
