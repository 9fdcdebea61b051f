"""Reduce the two rocprofv3 PMC passes of bench.py to profiles/rNN_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv \
        -- python3 bench.py --no-cpu-baseline --no-c5
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv \
        -- python3 bench.py --no-cpu-baseline --no-c5
    python scripts/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> NN

Per launch of the dominant kernel of each config (mean over the TIMED launches: the last four
dispatches of that kernel; the ones before are bench.py's untimed warm-up / pre-training).  Units and the gfx950 correction
follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE
count KiB; FETCH_SIZE reports half the bytes of wide coalesced reads, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import os
import sys

KERNELS = {'C3': 'k_tab_pwg', 'C2': 'k_tab_lpi<', 'C4': 'k_sr_wave<', 'C6': 'k_sfma'}
# bench.py times the LAST `--steps` (4) launches of each kernel; everything before them is untimed
# warm-up (C3: the pre-training that takes the agents to the full-work state, C6: 40 launches)
TIMED = 4


def per_kernel(path, counter):
    rows = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            if r['Counter_Name'] != counter:
                continue
            rows.setdefault(r['Kernel_Name'], []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


def slim(path, out, counter):
    """Keep one line per dispatch of the kernels that matter (the raw file is small already)."""
    with open(path, newline='') as fh, open(out, 'w', newline='') as oh:
        rd = csv.reader(fh)
        wr = csv.writer(oh)
        head = next(rd)
        wr.writerow(head)
        for row in rd:
            if row[head.index('Counter_Name')] == counter:
                wr.writerow(row)


def main():
    fetch_csv, write_csv, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
    fetch, write = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    out = {'_note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 '
                    'bench.py --no-cpu-baseline --no-c5` on MI355X, per launch of the dominant '
                    'kernel, mean over the four timed launches (the untimed warm-up / pre-training launches before them are dropped). '
                    'FETCH_SIZE and WRITE_SIZE are in KiB. Per MI355X_MICROARCH.md (HBM section) '
                    'FETCH_SIZE on gfx950 reports exactly half of the bytes of a wide coalesced '
                    'read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the table loads '
                    'of these kernels are 16-byte / 8-byte per lane coalesced streams, for which '
                    'that correction applies. Made by scripts/pmc_summary.py.'}
    for cfg, tag in KERNELS.items():
        names = [k for k in fetch if tag in k]
        assert len(names) == 1 and names[0] in write, (cfg, names)
        f, w = fetch[names[0]][-TIMED:], write[names[0]][-TIMED:]   # the timed launches
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        start = names[0].index(tag.rstrip('<'))
        short = names[0][start:names[0].index('(', start)]
        out[cfg] = {'kernel': short, 'launches': len(f), 'FETCH_SIZE_KiB': fk,
                    'WRITE_SIZE_KiB': wk, 'hbm_bytes_per_launch': (2 * fk + wk) * 1024}
    with open(os.path.join(root, 'r%02d_pmc_traffic.json' % rnd), 'w') as fh:
        json.dump(out, fh, indent=1)
    slim(fetch_csv, os.path.join(root, 'r%02d_pmc_fetch_size.csv' % rnd), 'FETCH_SIZE')
    slim(write_csv, os.path.join(root, 'r%02d_pmc_write_size.csv' % rnd), 'WRITE_SIZE')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
