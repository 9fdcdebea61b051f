"""Reduce the two rocprofv3 PMC passes of bench.py to profiles/rNN_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv \
        -- python3 bench.py --no-cpu-baseline --no-c5
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv \
        -- python3 bench.py --no-cpu-baseline --no-c5
    python scripts/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> NN

Per launch of the dominant kernel of each config (mean over the TIMED launches: the last four
dispatches of that kernel; the ones before are bench.py's untimed warm-up / pre-training).  Units and the gfx950 correction
follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE
count KiB; FETCH_SIZE reports half the bytes of wide coalesced reads.  The guide calibrates that
factor for streams only; scripts/pmc_calibrate_gather.py measured it on the access shapes of these
kernels' gathers (profiles/rNN_pmc_calibration.json: 2-, 4-, 8- and 16-byte per-lane loads, one
per 128-byte line of a 1 GiB table): every miss is ONE read request tallied at 64 bytes
(TCC_BUBBLE, the 128-byte request counter of FETCH_SIZE's formula, reads zero on gfx950) and fills
the whole 128-byte line (reading byte 64 of a line after byte 0 sends nothing to the fabric), so
the factor is 2 for every read width:
hbm_bytes = (factor * FETCH_SIZE + WRITE_SIZE) * 1024, factor from the calibration file.
"""
import glob
import csv
import json
import os
import sys

KERNELS = {'C3': 'k_tab_pwg', 'C2': 'k_tab_lpi<', 'C4': 'k_sr_wave<', 'C6': 'k_sfma',
           'general_hex_q': 'k_tab_wqn<true, 8,', 'general_wide_q': 'k_tab_wqn<true, 16,',
           'general_wide_q_lane': 'k_tab_general<'}
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


def calibrated_factor(root):
    """Line bytes per counted byte, the largest deviation from 2 over the calibrated read widths,
    and whether a miss fills the whole line (newest profiles/r*_pmc_calibration.json)."""
    files = sorted(glob.glob(os.path.join(root, 'r*_pmc_calibration.json')))
    if not files:
        return 2.0, None
    cal = json.load(open(files[-1]))
    fs = {k: e['factor_to_line_bytes'] for k, e in cal['patterns'].items() if 'factor_to_line_bytes' in e}
    whole = cal.get('halves', {}).get('read_requests_of_the_second_pass') == 0.0
    assert whole and all(abs(f - 2.0) < 0.01 for f in fs.values()), (fs, whole)
    return sum(fs.values()) / len(fs), {'file': os.path.basename(files[-1]), 'factor_by_pattern': fs,
                                        'a_miss_fills_the_whole_128_byte_line': whole}


def main():
    fetch_csv, write_csv, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
    factor, cal = calibrated_factor(root)
    fetch, write = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    out = {'_note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 '
                    'bench.py --no-cpu-baseline --no-c5` on MI355X, per launch of the dominant '
                    'kernel, mean over the four timed launches (the untimed warm-up / pre-training launches before them are dropped). '
                    'FETCH_SIZE and WRITE_SIZE are in KiB. hbm_bytes = (factor * FETCH_SIZE + '
                    'WRITE_SIZE) * 1024 with the factor CALIBRATED on these kernels\' access shapes '
                    '(`calibration`: scripts/pmc_calibrate_gather.py — coalesced 16-byte streams and '
                    '2 / 4 / 8 / 16-byte per-lane gathers each read as one 64-byte request per missed '
                    '128-byte line, and a miss fills the whole line). Made by scripts/pmc_summary.py.',
           'fetch_factor': factor, 'calibration': cal}
    for cfg, tag in KERNELS.items():
        names = [k for k in fetch if tag in k]
        if cfg.startswith('general_') and not names:
            continue          # (passes made without that leg)
        assert len(names) == 1 and names[0] in write, (cfg, names)
        f, w = fetch[names[0]][-TIMED:], write[names[0]][-TIMED:]   # the timed launches
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        start = names[0].index(tag.rstrip('<'))
        short = names[0][start:names[0].index('(', start)]
        out[cfg] = {'kernel': short, 'launches': len(f), 'FETCH_SIZE_KiB': fk,
                    'WRITE_SIZE_KiB': wk, 'fetch_factor': factor,
                    'hbm_bytes_per_launch': (factor * fk + wk) * 1024,
                    'hbm_bytes_per_launch_fetch_as_counted': (fk + wk) * 1024}
    path = os.path.join(root, 'r%02d_pmc_traffic.json' % rnd)
    if os.path.exists(path):   # (entries other reductions added — pmc_c5.py, pmc_fit.py — stay)
        for key, val in json.load(open(path)).items():
            out.setdefault(key, val)
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    slim(fetch_csv, os.path.join(root, 'r%02d_pmc_fetch_size.csv' % rnd), 'FETCH_SIZE')
    slim(write_csv, os.path.join(root, 'r%02d_pmc_write_size.csv' % rnd), 'WRITE_SIZE')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
