#!/bin/bash
# Reduce the counter passes of scripts/profile_pmc.sh / profile_pmc_rest.sh (merged back into
# gpurun_out/ by gpurun) into profiles/rNN_*.json:  bash scripts/reduce_pmc.sh 04
set -e
R=$1
N=$((10#$R))
O=gpurun_out
python scripts/pmc_sq.py $O/r${R}_sq_a.json $O/r${R}_sq_a/a_counter_collection.csv $O/r${R}_sq_b/b_counter_collection.csv $O/r${R}_sq_a/a_kernel_trace.csv $N > /dev/null
python scripts/pmc_summary.py $O/r${R}_pmc_fetch/f_counter_collection.csv $O/r${R}_pmc_write/w_counter_collection.csv $N > /dev/null
for dt in f64 f32; do
  python scripts/pmc_c5.py $O/r${R}_pmc_c5_fetch_$dt/f_counter_collection.csv $O/r${R}_pmc_c5_write_$dt/w_counter_collection.csv $N $dt > /dev/null
done
python scripts/pmc_fit.py $O/r${R}_pmc_fit_fetch/f_counter_collection.csv $O/r${R}_pmc_fit_write/w_counter_collection.csv $N > /dev/null
git status --short profiles | head
