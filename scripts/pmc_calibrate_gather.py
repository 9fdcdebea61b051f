"""Calibration of the rocprofv3 HBM counters on gather patterns.

MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streams (it reports half their
bytes on gfx950) and says "other access widths are uncalibrated: calibrate on a known byte count
in your own access pattern".  k_tab_pwg's global-memory waves gather 2-byte digest entries, random
16-byte Q rows and 4 / 8-byte records.  scripts/calib/gather_patterns.hip touches every 128-byte line
of a 1 GiB table exactly ONCE with loads of each of those widths (lanes of a load 64 lines apart or
more), streams the table with 16-byte loads, and — one workgroup — reads byte 0 and then byte 64 of
4 096 lines to tell whether a miss fills a whole line.

    python scripts/pmc_calibrate_gather.py build                    # hipcc, OUTSIDE any profiler
    python scripts/pmc_calibrate_gather.py run                      # the launches (under rocprofv3)
    python scripts/pmc_calibrate_gather.py reduce OUT.json DIR...   # counter CSVs -> per-pattern table

Passes (each its own process: MI355X_MICROARCH.md, PMC section):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d D1 -o c --output-format csv -- python3 scripts/pmc_calibrate_gather.py run
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d D2 ...
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d D3 ...
"""
import ctypes as C
import csv
import glob
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'calib', 'gather_patterns.hip')
LIB = os.path.join(HERE, 'calib', 'libpmc_calib.so')
TABLE_BYTES = 1 << 30
LINES = TABLE_BYTES // 128
PATTERNS = [  # (id, kernel name fragment, label, loads, bytes per load, distinct lines)
    (0, 'calib_stream16', '16 B per lane, coalesced stream of the table', TABLE_BYTES // 16, 16, LINES),
    (1, 'calib_touch<unsigned short>', '2 B per lane, one load per line (digest gathers)', LINES, 2, LINES),
    (2, 'calib_touch<unsigned int>', '4 B per lane, one load per line (model records)', LINES, 4, LINES),
    (3, 'rec8>', '8 B per lane, one load per line (digest rows)', LINES, 8, LINES),
    (4, 'row16>', '16 B per lane, one load per line (Q rows)', LINES, 16, LINES),
    (5, 'calib_halves', None, 4096, 4, 4096),
]


def stale():
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC)


def build():
    """Compile the calibration kernels.  Never called from `run`: under `rocprofv3 --pmc` the
    profiler's preloaded library has initialised the GPU before this script starts, and a compiler
    spawned from there is an exec from a GPU-initialised process (forbidden on this pool)."""
    if stale():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950',
                               '-fPIC', '-shared', '-o', LIB, SRC])


def run():
    if stale():
        sys.exit('scripts/calib/libpmc_calib.so is missing or older than gather_patterns.hip: run '
                 '`python3 scripts/pmc_calibrate_gather.py build` first, outside the profiler')
    import torch
    lib = C.CDLL(LIB)
    lib.calib_run.argtypes = [C.c_int, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_void_p]
    dev = torch.device('cuda', 0)
    table = torch.empty(TABLE_BYTES // 4, dtype=torch.int32, device=dev)
    table.random_(0, 1 << 30)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    scrub = torch.empty(768 << 20, dtype=torch.uint8, device=dev)   # > L2 + Infinity Cache
    st = torch.cuda.current_stream(dev).cuda_stream
    times = {}
    for p in (0, 1, 2, 3, 4):
        scrub.fill_(p)                      # nothing of the table is left in the caches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.calib_run(p, table.data_ptr(), TABLE_BYTES, sink.data_ptr(), st) == 0
        e1.record()
        torch.cuda.synchronize()
        times[p] = e0.elapsed_time(e1)
    # the two halves tests on a region of their own, in this order: first halves only, then both
    half = table.data_ptr() + (512 << 20)
    for p in (5, 6):
        scrub.fill_(p)
        assert lib.calib_run(p, half, 1 << 20, sink.data_ptr(), st) == 0
        torch.cuda.synchronize()
    print(json.dumps({'ms': times}))


def reduce_(out, dirs):
    counters = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path, newline='') as fh:
                for r in csv.DictReader(fh):
                    if 'calib_' not in r['Kernel_Name']:
                        continue
                    key = (r['Kernel_Name'], int(r['Dispatch_Id']))
                    counters.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    by_kernel = {}
    for (name, disp), c in sorted(counters.items(), key=lambda kv: kv[0][1]):
        by_kernel.setdefault(name, []).append(c)
    res = {'_note': 'scripts/pmc_calibrate_gather.py on one MI355X: every 128-byte line of a 1 GiB '
                    'table touched by exactly one load of the stated width (caches scrubbed before '
                    'each launch); FETCH_SIZE in KiB as rocprofv3 reports it; counted_bytes_per_line '
                    '= FETCH_SIZE * 1024 / distinct lines', 'patterns': {}}
    for pid, frag, label, loads, width, lines in PATTERNS:
        names = [k for k in by_kernel if frag in k]
        if not names:
            continue
        if pid == 5:
            one, both = by_kernel[names[0]][0], by_kernel[names[0]][1]
            rd1 = one.get('TCC_EA0_RDREQ_sum')
            rd2 = both.get('TCC_EA0_RDREQ_sum')
            res['halves'] = {
                'lines': lines, 'first_halves_only': one, 'both_halves': both,
                'read_requests_of_the_second_pass': None if rd1 is None else rd2 - rd1,
                'note': 'one workgroup reads byte 0 of 4 096 lines (first launch), and byte 0 then '
                        'byte 64 of them (second launch): requests the second pass adds = what '
                        'touching the other half of a resident line costs'}
            continue
        c = by_kernel[names[0]][0]
        e = {'label': label, 'loads': loads, 'bytes_per_load': width, 'distinct_lines': lines,
             'useful_bytes': loads * width, 'line_bytes': lines * 128, 'counters': c}
        if 'FETCH_SIZE' in c:
            e['FETCH_SIZE_bytes'] = c['FETCH_SIZE'] * 1024
            e['counted_bytes_per_line'] = c['FETCH_SIZE'] * 1024 / lines
            e['factor_to_line_bytes'] = lines * 128 / (c['FETCH_SIZE'] * 1024)
        if 'TCC_EA0_RDREQ_sum' in c:
            e['read_requests_per_line'] = c['TCC_EA0_RDREQ_sum'] / lines
        res['patterns'][frag] = e
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        run()
    else:
        reduce_(sys.argv[2], sys.argv[3:])
