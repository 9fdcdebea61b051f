#!/bin/bash
# The counter passes of a round on a GPU box (run through gpurun from the repository root):
#   bash scripts/profile_pmc.sh 04 [calib] [sq] [traffic]
# Every pass is its own process with --kernel-trace only (MI355X_MICROARCH.md, PMC section); the
# reductions (scripts/pmc_calibrate_gather.py reduce, pmc_sq.py, pmc_summary.py) run on the build
# machine afterwards and write profiles/rNN_*.json.
set -e
R=$1; shift
WHAT=${*:-calib sq traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
for w in $WHAT; do
 case $w in
 calib)
  python3 scripts/pmc_calibrate_gather.py build   # (hipcc outside the profiler: no exec from a GPU-initialised process)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_cal_fetch -o c --output-format csv -- python3 scripts/pmc_calibrate_gather.py run > $O/r${R}_cal_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $O/r${R}_cal_req -o c --output-format csv -- python3 scripts/pmc_calibrate_gather.py run > $O/r${R}_cal_req.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/r${R}_cal_hit -o c --output-format csv -- python3 scripts/pmc_calibrate_gather.py run > $O/r${R}_cal_hit.log 2>&1
  echo "calib done";;
 sq)
  CMD="python3 bench.py --full --no-cpu-baseline --min-seconds 0 --also C2,C6 --legs grid_search,general_hex_q,general_wide_q,general_dynaq_b100"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/r${R}_sq_a -o a --output-format csv -- $CMD > $O/r${R}_sq_a.json 2> $O/r${R}_sq_a.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/r${R}_sq_b -o b --output-format csv -- $CMD > $O/r${R}_sq_b.json 2> $O/r${R}_sq_b.err
  echo "sq done";;
 traffic)
  CMD="python3 bench.py --full --no-cpu-baseline --no-c5 --legs general_hex_q,general_wide_q,general_wide_q_lane --min-seconds 0"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_fetch -o f --output-format csv -- $CMD > $O/r${R}_pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_write -o w --output-format csv -- $CMD > $O/r${R}_pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $O/r${R}_pmc_req -o q --output-format csv -- $CMD > $O/r${R}_pmc_req.log 2>&1
  echo "traffic done";;
 esac
done
# (gpurun_out keeps <= 64 MiB: drop the raw traces that are not counter tables)
find $O -name "*_agent_info.csv" -delete 2>/dev/null || true
du -sh $O | tail -1
