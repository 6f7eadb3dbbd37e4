cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmcA -o a --output-format csv -- python $R/tools/bench_conv.py cfg4 bf16 decode5 > $R/gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmcB -o b --output-format csv -- python $R/tools/bench_conv.py cfg4 bf16 decode5 > $R/gpurun_out/pmcB.log 2>&1 &&
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA -d $R/gpurun_out/pmcC -o c --output-format csv -- python $R/tools/bench_conv.py cfg4 bf16 decode5 > $R/gpurun_out/pmcC.log 2>&1
ls -R $R/gpurun_out/pmcA | head
