export COBEL_DEBUG=1
mkdir -p gpurun_out
run() { # name, env..., args
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --full --also "" --min-seconds 0 --no-cpu-baseline $ARGS > gpurun_out/$name.json 2> gpurun_out/$name.err || { echo FAIL $name; tail -5 gpurun_out/$name.err; exit 1; }
  python -c "
import json,sys
r=json.load(open('gpurun_out/$name.json'))
print('$name', '%.4g'%r['value'], r['roofline']['launch_ms_all'])"
}
