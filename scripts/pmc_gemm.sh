set -x
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$R/tests/native/build/test_kernels
mkdir -p $R/gpurun_out/pmc
for shape in "fwd 3072 3072 768 0 1" "fwd 3072 768 3072 3 1" "fwd 4096 4096 4096 0 1" "fwd 8192 8192 8192 0 1" "dgrad 3072 768 3072 0 1" "wgrad 3072 3072 768 3 1"; do
  $T --one $shape 30
done > $R/gpurun_out/pmc/timing.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/a -- $T --one fwd 3072 3072 768 0 1 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/pmc/b -- $T --one fwd 3072 3072 768 0 1 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/c -- $T --one wgrad 3072 3072 768 3 1 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/d -- $T --one fwd 8192 8192 8192 0 1 3 > /dev/null 2>&1
find $R/gpurun_out/pmc -name "*.csv" | head
