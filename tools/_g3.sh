R=$PWD; cd /tmp; export TMPDIR=/tmp
for cfg in "256 256" "64 256 res"; do
tag=$(echo $cfg | tr ' ' '_')
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/p1_$tag -o pmc -- python $R/tools/_pmc1.py $cfg > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p2_$tag -o pmc -- python $R/tools/_pmc1.py $cfg > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p3_$tag -o pmc -- python $R/tools/_pmc1.py $cfg > /dev/null 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES TD_TD_BUSY TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCR_TCP_STALL_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/p4_$tag -o pmc -- python $R/tools/_pmc1.py $cfg > /dev/null 2>&1
done
cd $R
python tools/_pmcsum.py gpurun_out/p1_* gpurun_out/p2_* gpurun_out/p3_* gpurun_out/p4_* > gpurun_out/g3_pmc.txt 2>&1
find gpurun_out/p?_* -name "*.csv" -size +5M -delete
cat gpurun_out/g3_pmc.txt
