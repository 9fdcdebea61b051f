"""Reduce two rocprofv3 PMC passes over the C3 headline kernel to profiles/rNN_pmc_sq_c3.json.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
        SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/sq1 -o a \
        --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --also "" --min-seconds 0
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM \
        SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d gpurun_out/sq2 -o b --output-format csv -- (same)
    python scripts/pmc_sq.py <a_counter_collection.csv> <b_counter_collection.csv> NN [kernel tag]

Mean over the four TIMED launches (the last four dispatches of the kernel: bench.py's pre-training
launches come before them)."""
import csv
import json
import os
import sys


def main():
    rnd = int(sys.argv[3])
    tag = sys.argv[4] if len(sys.argv) > 4 else 'k_tab_pwg'
    acc = {}
    for path in sys.argv[1:3]:
        rows = {}
        with open(path, newline='') as fh:
            for r in csv.DictReader(fh):
                if tag in r['Kernel_Name']:
                    rows.setdefault(r['Counter_Name'], {}).setdefault(int(r['Dispatch_Id']), 0.0)
                    rows[r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
        for name, by_dispatch in rows.items():
            vals = [v for _, v in sorted(by_dispatch.items())][-4:]
            acc[name] = sum(vals) / len(vals)
    steps = 65536 * 512
    out = dict(acc)
    out['_per_env_step'] = {k: round(acc[k] / steps, 2) for k in acc if k.startswith('SQ_INSTS_')}
    if 'SQ_WAVE_CYCLES' in acc:
        out['_shares_of_wave_cycles'] = {k: round(acc[k] / acc['SQ_WAVE_CYCLES'], 3) for k in acc
                                         if k.startswith('SQ_ACTIVE_') or k.startswith('SQ_WAIT')}
    out['_note'] = ('rocprofv3 --kernel-trace --pmc (two passes) of `python3 bench.py --no-cpu-baseline '
                    '--no-c5 --also "" --min-seconds 0`: C3 headline kernel %s, mean over the four timed '
                    'launches (trained agents, >= 95 %% of the planning batches evaluated), 65536 '
                    'instances x 512 env steps per launch' % tag)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles',
                        'r%02d_pmc_sq_c3.json' % rnd)
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
