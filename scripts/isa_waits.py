"""Scan the library's code objects (no GPU needed) for memory requests that are waited for right
behind their issue INSIDE a loop — the signature of a prefetch the register allocator turned into
a stall (a loaded tuple copied out behind `s_waitcnt vmcnt`, a spilled load, a load under a
condition joined with others): round 5 found such a wait costing 12 % of a step in two builds of
k_tab_pwg (scripts/experiments/pwg_r05/README.md).

    python scripts/isa_waits.py [file.hip ...] [--window 6]

Prints, per kernel, every vector-memory load in a loop that is followed within `window`
instructions by an `s_waitcnt vmcnt(N)` small enough to cover it (N <= loads issued since).
Immediate consumption is sometimes intended (a dependent gather): read the listing, then the ISA.
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_census import CSRC, FLAGS, demangle, sources  # noqa: E402


def scan(asm_path, window):
    lines = open(asm_path).read().split('\n')
    out, kernel, depth = [], None, 0
    names = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kernel = m.group(1)
        if l.startswith('.Lfunc_end'):
            kernel = None
        if kernel is None:
            continue
        m = re.search(r'Loop Header: Depth=(\d+)|in Loop: Header=\S+ Depth=(\d+)', l)
        if m:
            depth = int(m.group(1) or m.group(2))
        elif re.match(r'^\.LBB\d+_\d+:\s*$', l) or re.match(r'^; %bb\.\d+:\s*$', l):
            depth = 0
        if depth == 0 or not re.match(r'\t(global_load|buffer_load|flat_load|scratch_load)', l):
            continue
        later_loads = 0
        for j in range(i + 1, min(len(lines), i + 1 + window * 2)):
            t = lines[j].strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            if re.match(r'(global_load|buffer_load|flat_load|scratch_load)', t):
                later_loads += 1
            w = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
            if w and int(w.group(1)) <= later_loads:
                out.append((kernel, depth, i + 1, l.strip(), t))
                break
            window_left = window - (j - i)
            if window_left <= 0:
                break
        names[kernel] = None
    return out


def dpp_hazards(asm_path):
    """The hand-written `v_max_f32_dpp` blocks (tabular.hip, tabular_pwg.hip, tabular_nact.hip) place
    their own wait states: the compiler's hazard recognizer does not look inside an asm block.  What
    the blocks cover is a VALU write of the SOURCE register (two wait states: `s_nop 1`).  What they
    do not: a VALU write of EXEC (`v_cmpx_*`) within five wait states ahead of a DPP instruction.
    Reports every DPP instruction with a `v_cmpx` closer than that in the shipped code."""
    lines = open(asm_path).read().split('\n')
    out, kernel = [], None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kernel = m.group(1)
        if kernel is None or '_dpp' not in l or not re.match(r'\tv_', l):
            continue
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            t = lines[j].strip()
            j -= 1
            if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                continue
            if t.startswith('v_cmpx'):
                out.append((kernel, i + 1, l.strip(), t, states))
                break
            nop = re.match(r's_nop (\d+)', t)
            states += int(nop.group(1)) + 1 if nop else 1
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    window = 6
    if '--window' in sys.argv:
        window = int(sys.argv[sys.argv.index('--window') + 1])
        args = [a for a in args if a != str(window)]
    files = args or sources()
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(CSRC, f)
        with tempfile.TemporaryDirectory() as tmp:
            asm = os.path.join(tmp, 'k.s')
            subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-S', '--cuda-device-only', '-o', asm, src],
                           check=True, capture_output=True)
            hits = scan(asm, window)
            for k, line, dpp, cmpx, st in dpp_hazards(asm):
                print('%-12s DPP HAZARD line %d: %s only %d wait states behind %s' % (
                    os.path.basename(f)[:-4], line, dpp, st, cmpx))
        if not hits:
            continue
        dm = demangle(sorted({h[0] for h in hits}))
        for k, depth, line, load, wait in hits:
            name = dm.get(k, k).replace('(anonymous namespace)::', '')[:70]
            print('%-12s %-70s depth %d line %5d  %-48s -> %s' % (os.path.basename(f)[:-4], name, depth, line, load[:48], wait))


if __name__ == '__main__':
    main()
