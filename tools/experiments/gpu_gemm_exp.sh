# per-kernel times of the layer-by-layer step with the GEMM's k-loop alone (gx1) / epilogue alone (gx2): where a GEMM launch spends its time
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04w_exp; mkdir -p $O
for v in ${VARIANTS:-gx1 gx2}; do
  export INRFIT_LIB=variants/libinrfit_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/st_$v -o w -- python3 tools/kbench_wide.py ${SHAPES:-256x1} > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  db=$(find $O/st_$v -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" $O/${v}_kernel_stats.csv; else cp $(find $O/st_$v -name "*kernel_stats.csv" | head -1) $O/${v}_kernel_stats.csv; fi
  rm -rf $O/st_$v
  echo "== $v"; grep "gemm_kernel" $O/${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c30-120
done
if [ -n "$PMC_VARIANT" ]; then
  export INRFIT_LIB=variants/libinrfit_$PMC_VARIANT.so
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $O/pmc -o p -- python3 tools/kbench_wide.py ${SHAPES:-256x1} > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
  db=$(find $O/pmc -name "*.db" | head -1); csv=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$csv" ]; then python3 tools/pmc_summary.py "$csv" > $O/pmc_$PMC_VARIANT.txt; else python3 tools/pmc_summary_db.py "$db" > $O/pmc_$PMC_VARIANT.txt; fi
  rm -rf $O/pmc
  grep -A9 "gemm_kernel" $O/pmc_$PMC_VARIANT.txt | head -40
fi
