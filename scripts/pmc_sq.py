"""Reduce two rocprofv3 SQ counter passes of bench.py to profiles/rNN_pmc_sq.json: per leg whose
kernel is bound by instruction issue, the wave-level instructions per env step and how busy the
vector ALUs were — what bench.py's `roofline.issue` objects are made of.

    A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
    B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
    CMD="python3 bench.py --full --no-cpu-baseline --min-seconds 0 --also C2,C6 --legs grid_search,general_dynaq_b100,general_hex_q,general_wide_q"
    rocprofv3 --kernel-trace --pmc $A -d gpurun_out/sq_a -o a --output-format csv -- $CMD > gpurun_out/sq_a.json
    rocprofv3 --kernel-trace --pmc $B -d gpurun_out/sq_b -o b --output-format csv -- $CMD > gpurun_out/sq_b.json
    python scripts/pmc_sq.py gpurun_out/sq_a.json <a_counter_collection.csv> <b_counter_collection.csv> <a_kernel_trace.csv> NN

Per leg: the TIMED launches of its kernel (the last four dispatches; everything before them is
bench.py's untimed warm-up / pre-training), for the grid search every dispatch of its one training
session.  Counters are in units of four cycles (one wave64 vector instruction); `valu_busy_frac` =
SQ_ACTIVE_INST_VALU / (1024 SIMDs x the dispatch's duration in four-cycle units at the device clock
the kernel trace implies: duration x clock, clock taken from GRBM_GUI_ACTIVE / duration when that
counter is there, else 2.4 GHz)."""
import csv
import json
import os
import sys

LEGS = {   # leg -> (kernel name fragment, what must not be in the name, timed dispatches or None = all)
    'C3': ('k_tab_pwg', None, 4),
    'C2': ('k_tab_lpi<', None, 4),
    'C6': ('k_sfma', None, 4),
    'grid_search': ('k_tab_wpi<', None, None),
    'general_dynaq_b100': ('k_tab_wpi<', None, 4),
    'general_hex_q': ('k_tab_wqn<true, 8,', None, 4),
    'general_wide_q': ('k_tab_wqn<true, 16,', None, 4),
}


def load(path):
    rows = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            d = rows.setdefault(int(r['Dispatch_Id']), {'kernel': r['Kernel_Name']})
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return rows


def durations(path):
    out = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            out[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    return out


def main():
    bench = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith('{')][-1])
    a, b, dur = load(sys.argv[2]), load(sys.argv[3]), durations(sys.argv[4])
    rnd = int(sys.argv[5])
    legs = bench.get('other_configs', {})
    steps = {'C3': bench['config']['instances_per_gpu'] * bench['config']['env_steps_per_launch']}
    for k in ('C2', 'C6'):
        if k in legs and 'config' in legs[k]:
            steps[k] = legs[k]['config']['instances_per_gpu'] * legs[k]['config']['env_steps_per_launch']
    for k in ('general_dynaq_b100', 'general_hex_q', 'general_wide_q'):
        if k in legs and 'config' in legs[k]:
            c = legs[k]['config']
            steps[k] = c['instances_per_gpu'] * c['env_steps_per_launch']
    out = {'_note': 'rocprofv3 --kernel-trace --pmc (two passes, scripts/pmc_sq.py) of bench.py on one '
                    'MI355X; per leg the timed launches of its kernel; instruction counts are '
                    'wave-level (one per wavefront instruction), per env step of the leg'}

    for leg, (frag, _, timed) in LEGS.items():
        if leg not in steps and leg != 'grid_search':
            continue
        ida = [d for d in sorted(a) if frag in a[d]['kernel']]
        idb = [d for d in sorted(b) if frag in b[d]['kernel']]
        if leg in ('grid_search', 'general_dynaq_b100'):
            # both legs run k_tab_wpi: the grid search's launches come first (its parameter-set
            # instantiation), the B = 100 leg's pre-training runs k_tab_pwg and its timed launches
            # are the LAST four k_tab_wpi dispatches of the run
            if leg == 'general_dynaq_b100':
                ida, idb = ida[-4:], idb[-4:]
            else:
                name = a[ida[0]]['kernel'] if ida else ''
                ida = [d for d in ida if a[d]['kernel'] == name]
                idb = [d for d in idb if b[d]['kernel'] == name]
        elif timed:
            if leg == 'C3':
                # the headline leg comes first: pre-training + the timed launches; the B = 100 leg,
                # much later in the run, pre-trains with 48 more launches of this kernel — the
                # headline's dispatches end at the first large gap in the dispatch ids
                def head(ids):
                    for k in range(1, len(ids)):
                        if ids[k] - ids[k - 1] > 1000:
                            return ids[:k]
                    return ids
                ida, idb = head(ida), head(idb)
            ida, idb = ida[-timed:], idb[-timed:]
        if not ida or not idb:
            continue
        tot_a = {k: sum(a[d].get(k, 0.0) for d in ida) for k in a[ida[0]] if k != 'kernel'}
        tot_b = {k: sum(b[d].get(k, 0.0) for d in idb) for k in b[idb[0]] if k != 'kernel'}
        n_steps = (legs['grid_search']['env_steps'] if leg == 'grid_search'
                   else steps[leg] * len(idb))
        secs = sum(dur[d] for d in ida)
        clock = 2.4e9
        if tot_a.get('GRBM_GUI_ACTIVE'):
            # (summed over the eight XCCs by the counter service)
            clock = tot_a['GRBM_GUI_ACTIVE'] / 8.0 / secs
            if not 1.0e9 < clock < 3.0e9:
                clock = 2.4e9
        e = {'kernel': a[ida[0]]['kernel'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:80], 'dispatches': len(ida),
             'env_steps': n_steps, 'seconds': secs, 'clock_hz_assumed': clock,
             'valu_per_env_step': tot_b['SQ_INSTS_VALU'] / n_steps,
             'salu_per_env_step': tot_b['SQ_INSTS_SALU'] / n_steps,
             'lds_per_env_step': tot_b['SQ_INSTS_LDS'] / n_steps,
             'vmem_per_env_step': (tot_b['SQ_INSTS_VMEM_RD'] + tot_b['SQ_INSTS_VMEM_WR']) / n_steps,
             'valu_busy_frac': tot_a['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024 * clock * secs),
             'shares_of_wave_cycles': {k: round(tot_a[k] / tot_a['SQ_WAVE_CYCLES'], 3) for k in tot_a
                                       if k.startswith('SQ_ACTIVE_') or k.startswith('SQ_WAIT')},
             'raw': {**tot_a, **tot_b}}
        out[leg] = e
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles',
                        'r%02d_pmc_sq.json' % rnd)
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'raw'} if isinstance(v, dict) else v
                      for k, v in out.items()}, indent=1))


if __name__ == '__main__':
    main()
