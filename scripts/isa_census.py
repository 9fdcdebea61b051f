"""Census of the library's code objects (no GPU needed): every kernel's registers, scratch and
spills from the metadata hipcc writes, and the instructions that tell of trouble inside loops —
scratch reloads, scalar-spill moves (v_readlane / v_writelane), float64 divisions.

    python scripts/isa_census.py [file.hip ...]        # default: every source of csrc/Makefile

Three of round 3's largest gains were read off this output, not off a profile (DESIGN.md §4.2 iv/v,
§4.2c): `private_segment_fixed_size` > 0 in instantiations that should have been register-only, a
`scratch_load` + `s_waitcnt vmcnt(0)` in front of a gather, a `v_div_scale_f64` sequence for a
switch the kernel is never launched with.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cobel-rl_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC',
         '-fvisibility=hidden', '-munsafe-fp-atomics', '-I' + os.path.join(ROOT, 'include')]


def sources():
    mk = open(os.path.join(CSRC, 'Makefile')).read()
    return re.search(r'^SRCS\s*:=\s*(.*)$', mk, re.M).group(1).split()


def demangle(names):
    for tool in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', 'c++filt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True,
                                 check=True).stdout.split('\n')
            return dict(zip(names, out))
        except Exception:
            continue
    return {n: n for n in names}


def census(asm_path):
    text = open(asm_path).read().split('\n')
    # kernel bodies: from the label to s_endpgm of the last block (next label of a kernel / .section)
    meta = {}
    joined = '\n'.join(text)
    kernels = joined[joined.index('amdhsa.kernels:'):] if 'amdhsa.kernels:' in joined else ''
    for blk in re.split(r'\n  - \.agpr_count:', kernels)[1:]:
        name = re.search(r'\n    \.name:\s+(\S+)', blk).group(1)
        meta[name] = {k: int(re.search(r'\n    \.%s:\s+(\d+)' % k, blk).group(1))
                      for k in ('vgpr_count', 'sgpr_spill_count', 'vgpr_spill_count',
                                'private_segment_fixed_size')}
    rows = []
    for name, md in meta.items():
        start = next((k for k, l in enumerate(text) if l.startswith(name + ':')), None)
        if start is None:
            continue
        end = start
        while end < len(text) and not text[end].startswith('.Lfunc_end'):
            end += 1
        body = text[start:end]
        in_loop = [('in Loop:' in l) or ('Loop Header' in l) for l in body]
        loop_recent, last = [], -10 ** 9
        for k, flag in enumerate(in_loop):
            if flag:
                last = k
            loop_recent.append(k - last < 80)
        count = lambda pat, only_loops: sum(1 for k, l in enumerate(body)
                                            if re.search(pat, l) and (loop_recent[k] or not only_loops))
        rows.append(dict(name=name, lines=len(body), **md,
                         scratch_in_loops=count(r'\bscratch_load', True),
                         spill_moves_in_loops=count(r'v_readlane_b32|v_writelane_b32', True),
                         f64_divisions=count(r'v_div_scale_f64', False) // 2))
    return rows


def main():
    files = [a for a in sys.argv[1:] if a.endswith('.hip')] or sources()
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in files:
            src = f if os.path.isabs(f) else os.path.join(CSRC, f)
            stem = os.path.splitext(os.path.basename(src))[0]
            procs.append((stem, subprocess.Popen(
                ['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', src, '-o', os.path.join(tmp, stem + '.o'),
                                                   '-save-temps=obj'],
                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)))
        rows = []
        for stem, p in procs:
            assert p.wait() == 0, 'hipcc failed on ' + stem
            for r in census(os.path.join(tmp, stem + '-hip-amdgcn-amd-amdhsa-gfx950.s')):
                rows.append(dict(r, file=stem))
    names = demangle([r['name'] for r in rows])
    print('%-12s %5s %5s %6s %6s %7s %8s %7s %6s  %s' % (
        'file', 'lines', 'vgpr', 'v-spil', 's-spil', 'scratch', 'scr@loop', 'rl/wl@l', 'f64div', 'kernel'))
    for r in sorted(rows, key=lambda r: (-r['private_segment_fixed_size'], -r['scratch_in_loops'],
                                          -r['sgpr_spill_count'])):
        short = re.sub(r'\(anonymous namespace\)::', '', names[r['name']])
        short = re.sub(r'\(.*\)$', '', short)
        print('%-12s %5d %5d %6d %6d %7d %8d %7d %6d  %s' % (
            r['file'], r['lines'], r['vgpr_count'], r['vgpr_spill_count'], r['sgpr_spill_count'],
            r['private_segment_fixed_size'], r['scratch_in_loops'], r['spill_moves_in_loops'],
            r['f64_divisions'], short[:90]))


if __name__ == '__main__':
    main()
