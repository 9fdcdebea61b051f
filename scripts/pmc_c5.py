"""Add the DQN replay kernel's HBM traffic to profiles/rNN_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_c5_fetch -o f --output-format csv \
        -- python3 scripts/run_c5.py f64 64
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_c5_write -o w --output-format csv \
        -- python3 scripts/run_c5.py f64 64
    python scripts/pmc_c5.py <fetch counter_collection.csv> <write counter_collection.csv> NN f64

Same units and gfx950 correction as scripts/pmc_summary.py (FETCH_SIZE doubled: the parameter and
moment streams are coalesced 8- / 32-byte-per-lane reads).  Mean per launch over all dispatches of
k_dqn_replay_lds but the first eight (warm-up run)."""
import json
import os
import sys

from pmc_summary import per_kernel


def main():
    fetch_csv, write_csv, rnd, dt = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles',
                        'r%02d_pmc_traffic.json' % rnd)
    out = json.load(open(path))
    fetch, write = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    tag = 'k_dqn_replay_lds_%s<' % dt
    name = [k for k in fetch if tag in k]
    assert len(name) == 1 and name[0] in write, name
    f, w = fetch[name[0]][8:], write[name[0]][8:]
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out['C5_' + dt] = {'kernel': tag, 'launches': len(f), 'FETCH_SIZE_KiB': fk,
                       'WRITE_SIZE_KiB': wk, 'hbm_bytes_per_launch': (2 * fk + wk) * 1024,
                       'command': 'python3 scripts/run_c5.py %s 64' % dt}
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out['C5_' + dt], indent=1))


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
