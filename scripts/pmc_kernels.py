"""Per-kernel means of rocprofv3 --pmc passes (any number of counter_collection.csv files of the
SAME command) next to the kernel-trace durations: a quick table for kernels that have no leg in
pmc_sq.py yet.

    python scripts/pmc_kernels.py <kernel_trace.csv> <counter_collection.csv>... [--match frag,frag]

Counters of the SQ block are in units of four cycles per SIMD (one wave64 vector instruction) unless
their name says otherwise; the `busy` columns divide by 1024 SIMDs x the launch's duration in those
units at 2.4 GHz."""
import csv
import sys
from collections import defaultdict

SIMDS, CLOCK = 1024, 2.4e9


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--match')]
    match = None
    for k, a in enumerate(sys.argv):
        if a == '--match':
            match = sys.argv[k + 1].split(',')
            args = [x for x in args if x != sys.argv[k + 1]]
    trace, passes = args[0], args[1:]
    dur = defaultdict(list)
    with open(trace, newline='') as fh:
        for r in csv.DictReader(fh):
            dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
    ctr = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
    pdur = defaultdict(lambda: defaultdict(dict))
    for p in passes:
        with open(p, newline='') as fh:
            for r in csv.DictReader(fh):
                ctr[r['Kernel_Name']][r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
    for name in sorted(dur, key=lambda n: -sum(dur[n])):
        if match and not any(m in name for m in match):
            continue
        if name not in ctr:
            continue
        d = dur[name]
        # (steady state: the median launch)
        med = sorted(d)[len(d) // 2]
        print('%s\n  launches %d, median %.1f us (trace of the first pass)' % (name[:110], len(d), med * 1e6))
        slots = SIMDS * med * CLOCK / 4.0
        for c in sorted(ctr[name]):
            vals = sorted(ctr[name][c].values())
            v = vals[len(vals) // 2]
            extra = ''
            if c.startswith('SQ_ACTIVE_INST') or c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU',
                                                          'SQ_INSTS_MFMA', 'SQ_WAIT_INST_ANY',
                                                          'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY',
                                                          'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_LDS'):
                extra = '   / SIMD slots = %.3f' % (v / slots)
            print('    %-28s %14.4g%s' % (c, v, extra))


if __name__ == '__main__':
    main()
