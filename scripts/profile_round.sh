#!/bin/bash
# One round's profiles on a GPU box (run through gpurun from the repository root):
#   bash scripts/profile_round.sh 04
# 1. bench.py as the driver runs it                   -> gpurun_out/rNN_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same run  -> gpurun_out/rNN_stats/ (+ the timed launches
#    of each headline kernel: scripts/kernel_stats_timed.py)
# 3. the counter passes (scripts/profile_pmc.sh: calibration, SQ, traffic; every pass its own
#    process, --kernel-trace only — MI355X_MICROARCH.md) and those of the C5 / Dyna-DSR fit legs
# The summaries are reduced and copied into profiles/ afterwards on the build machine
# (scripts/pmc_calibrate_gather.py reduce, pmc_sq.py, pmc_summary.py, pmc_c5.py): profiles/ is
# tracked, gpurun_out/ is scratch.
set -e
R=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
python3 bench.py --full-out $O/r${R}_bench_full.json > $O/r${R}_bench.json 2> $O/r${R}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/r${R}_stats -o s --output-format csv -- python3 bench.py --full --no-cpu-baseline --min-seconds 0 > $O/r${R}_stats.log 2>&1
python3 scripts/kernel_stats_timed.py $O/r${R}_stats/s_kernel_trace.csv $O/r${R}_kernel_stats_timed.csv > /dev/null
echo "stats done"
bash scripts/profile_pmc.sh $R calib sq traffic
for dt in f64 f32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_c5_fetch_$dt -o f --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_fetch_$dt.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_c5_write_$dt -o w --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_write_$dt.log 2>&1
done
echo "c5 pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_fit_fetch -o f --output-format csv -- python3 scripts/experiments/exp_mlp_fit.py > $O/r${R}_pmc_fit_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_fit_write -o w --output-format csv -- python3 scripts/experiments/exp_mlp_fit.py > $O/r${R}_pmc_fit_write.log 2>&1
echo "fit pmc done"
# keep what the reductions read; the kernel traces of the counter passes are large
find $O -name "*_agent_info.csv" -delete 2>/dev/null || true
rm -f $O/r${R}_pmc_fetch/f_kernel_trace.csv $O/r${R}_pmc_write/w_kernel_trace.csv $O/r${R}_pmc_req/q_kernel_trace.csv $O/r${R}_sq_b/b_kernel_trace.csv
du -sh $O | tail -1
