"""From a rocprofv3 kernel trace of `python3 bench.py --no-cpu-baseline --min-seconds 0`: the
durations of the launches bench.py TIMES for each headline kernel — the last `--steps` (4)
dispatches of that kernel inside its leg; the dispatches before them are the untimed warm-up /
pre-training launches (C3 pre-trains for dozens of launches on younger, cheaper agents, so the
per-kernel average of rocprofv3's own kernel_stats.csv is not the timed launches' average).

    python scripts/kernel_stats_timed.py <s_kernel_trace.csv> <out.csv>
"""
import csv
import sys

# kernel-name tag -> (config, dispatches of that kernel that belong to the leg's timed window)
LEGS = [('k_tab_pwg', 'C3', 4), ('k_tab_lpi<', 'C2', 4), ('k_sr_wave<', 'C4', 4), ('k_sfma', 'C6', 4)]


def main():
    trace, out = sys.argv[1], sys.argv[2]
    rows = {}
    with open(trace, newline='') as fh:
        for r in csv.DictReader(fh):
            rows.setdefault(r['Kernel_Name'], []).append(
                (int(r['Dispatch_Id']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    with open(out, 'w', newline='') as oh:
        wr = csv.writer(oh)
        wr.writerow(['Config', 'Name', 'Calls_in_trace', 'Timed_calls', 'Timed_AverageNs',
                     'Timed_MinNs', 'Timed_MaxNs', 'All_AverageNs'])
        for tag, cfg, timed in LEGS:
            names = [k for k in rows if tag in k]
            if cfg == 'C3' and len(names) > 1:      # (the B = 100 leg also runs C3's kernel briefly)
                names = sorted(names, key=lambda k: -len(rows[k]))[:1]
            for name in names:
                d = [x for _, x in sorted(rows[name])]
                if cfg == 'C3':
                    # the headline leg comes first in the run: pre-training + 4 timed launches, then
                    # (in the B = 100 leg, much later) 48 more pre-training launches of this kernel.
                    # The headline's timed window = the 4 dispatches before the largest gap in ids.
                    ids = [i for i, _ in sorted(rows[name])]
                    cut = len(ids)
                    for k in range(1, len(ids)):
                        if ids[k] - ids[k - 1] > 1000:
                            cut = k
                            break
                    d = d[:cut]
                t = d[-timed:]
                wr.writerow([cfg, name, len(d), len(t), sum(t) / len(t), min(t), max(t),
                             sum(d) / len(d)])
    print(open(out).read())


if __name__ == '__main__':
    main()
