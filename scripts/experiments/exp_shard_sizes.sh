# Launch time of C3 shards (what each GPU of an N-way split runs), driver-style windows:
#   bash scripts/experiments/exp_shard_sizes.sh     (through gpurun, from the repository root)
source scripts/ab_pwg.sh
for n in 65536 32768 16384 8192; do
  ARGS="--instances $n"
  run shard_$n A=1
done
