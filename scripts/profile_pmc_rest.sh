#!/bin/bash
# The counter passes of the network legs (C5 replay kernel, Dyna-DSR fit launch): part of
# scripts/profile_round.sh, callable on its own:  bash scripts/profile_pmc_rest.sh 04
set -e
R=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
for dt in f64 f32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_c5_fetch_$dt -o f --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_fetch_$dt.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_c5_write_$dt -o w --output-format csv -- python3 scripts/run_c5.py $dt 64 > $O/r${R}_pmc_c5_write_$dt.log 2>&1
done
echo "c5 pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r${R}_pmc_fit_fetch -o f --output-format csv -- python3 scripts/experiments/exp_mlp_fit.py > $O/r${R}_pmc_fit_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r${R}_pmc_fit_write -o w --output-format csv -- python3 scripts/experiments/exp_mlp_fit.py > $O/r${R}_pmc_fit_write.log 2>&1
echo "fit pmc done"
find $O -name "*_agent_info.csv" -delete 2>/dev/null || true
