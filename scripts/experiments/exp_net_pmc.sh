cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
CMD="python3 bench.py --full --no-cpu-baseline --min-seconds 0 --max-pretrain 2 --also= --legs C5_f32,C5_f64,dyna_dqn,dyna_dsr"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA -d $O/net_a -o a --output-format csv -- $CMD > $O/net_a.json 2> $O/net_a.err &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/net_b -o b --output-format csv -- $CMD > $O/net_b.json 2> $O/net_b.err &&
python3 scripts/pmc_kernels.py $O/net_a/a_kernel_trace.csv $O/net_a/a_counter_collection.csv $O/net_b/b_counter_collection.csv --match dqn_replay,mlp_fit,mlp_forward,dqn_act,dqn_batch,dsr_targets > $O/net_pmc.txt
rm -f $O/net_a/a_agent_info.csv $O/net_b/b_agent_info.csv
tail -c 3000 $O/net_a.err; du -sh $O/net_a $O/net_b
