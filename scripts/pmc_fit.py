"""Add the Dyna-DSR training kernel's HBM traffic to profiles/rNN_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fit_fetch -o f --output-format csv \
        -- python3 scripts/experiments/exp_mlp_fit.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_fit_write -o w --output-format csv \
        -- python3 scripts/experiments/exp_mlp_fit.py
    python scripts/pmc_fit.py <fetch counter_collection.csv> <write counter_collection.csv> NN

scripts/experiments/exp_mlp_fit.py launches cobel_mlp_fit on 32 768 float64 25-64-64-25 networks (the successor
networks of 8 192 Dyna-DSR agents): its first six launches are the full step (the later ones leave
parts out).  Same units as scripts/pmc_summary.py; FETCH_SIZE is reported both as counted and with
the guide's doubling for wide coalesced reads (these are 8- and 16-byte-per-lane loads)."""
import json
import os
import sys

from pmc_summary import per_kernel


def main():
    fetch_csv, write_csv, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles',
                        'r%02d_pmc_traffic.json' % rnd)
    out = json.load(open(path))
    fetch, write = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    name = [k for k in fetch if 'k_mlp_fit<double>' in k]
    assert len(name) == 1 and name[0] in write, list(fetch)
    f, w = fetch[name[0]][:6], write[name[0]][:6]
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    n, params = 32768, 64 * 25 + 64 + 64 * 64 + 64 + 25 * 64 + 25
    out['dyna_dsr_fit'] = {'kernel': 'k_mlp_fit<double>', 'launches': len(f), 'networks': n,
                           'FETCH_SIZE_KiB': fk, 'WRITE_SIZE_KiB': wk,
                           'hbm_bytes_per_launch': (2 * fk + wk) * 1024,
                           'hbm_bytes_per_launch_fetch_as_counted': (fk + wk) * 1024,
                           'algorithmic_bytes_per_launch': n * 8 * params * 8,
                           'command': 'python3 scripts/experiments/exp_mlp_fit.py'}
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out['dyna_dsr_fit'], indent=1))


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
