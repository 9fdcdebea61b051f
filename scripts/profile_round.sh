#!/bin/bash
# One round's profiles on a GPU box (run through gpurun from the repository root):
#   bash scripts/profile_round.sh 02
# 1. bench.py as the driver runs it                   -> gpurun_out/rNN_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same run  -> gpurun_out/rNN_stats/
# 3. PMC passes (FETCH_SIZE, WRITE_SIZE separately, kernel-trace only — MI355X_MICROARCH.md)
#    of bench.py's tabular legs and of the C5 leg      -> gpurun_out/rNN_pmc_*/
# The summaries are reduced and copied into profiles/ by scripts/pmc_summary.py / pmc_c5.py
# afterwards (on the build machine: profiles/ is tracked, gpurun_out/ is scratch).
set -e
R=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/r${R}_bench.json 2> $O/r${R}_bench.err
echo "bench done"
# (--min-seconds 0 --no-pretrain-timing is not needed: the C3 kernel's pre-training launches run
#  the same kernel on younger agents, so the per-kernel average of the trace is LOWER than the timed
#  launches'; scripts/kernel_stats_timed.py extracts the timed launches from the kernel trace)
rocprofv3 --kernel-trace --stats -d $O/r${R}_stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --min-seconds 0 > $O/r${R}_stats.log 2>&1
python3 scripts/kernel_stats_timed.py $O/r${R}_stats/s_kernel_trace.csv $O/r${R}_kernel_stats_timed.csv > /dev/null
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --min-seconds 0 > $O/r${R}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --min-seconds 0 > $O/r${R}_pmc_write.log 2>&1
echo "write done"
for dt in f64 f32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_c5_fetch_$dt -o f --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_fetch_$dt.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_c5_write_$dt -o w --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_write_$dt.log 2>&1
done
echo "c5 pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_fit_fetch -o f --output-format csv -- python3 scripts/exp_mlp_fit.py > $O/r${R}_pmc_fit_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_fit_write -o w --output-format csv -- python3 scripts/exp_mlp_fit.py > $O/r${R}_pmc_fit_write.log 2>&1
echo "fit pmc done"
