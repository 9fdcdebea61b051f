cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/fit_sq1 -o a --output-format csv -- python3 scripts/exp_mlp_fit.py > gpurun_out/fit_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA -d gpurun_out/fit_sq2 -o b --output-format csv -- python3 scripts/exp_mlp_fit.py > gpurun_out/fit_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/fit_sq3 -o c --output-format csv -- python3 scripts/exp_mlp_fit.py > gpurun_out/fit_sq3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr -d gpurun_out/fit_sq4 -o d --output-format csv -- python3 scripts/exp_mlp_fit.py > gpurun_out/fit_sq4.log 2>&1
tail -2 gpurun_out/fit_sq4.log
