"""Static instruction census of the step loops of k_tab_pwg (no GPU): per depth-2 loop of the
<false, true> instantiation (the headline's), the instructions by unit in the loop's text — rare
blocks included, so the numbers compare builds, they are not per-step counts.
    python scripts/isa_loop_count.py [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_census import CSRC, FLAGS  # noqa: E402

def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + sys.argv[1:] + ['-S', '--cuda-device-only', '-o', asm,
                        os.path.join(CSRC, 'tabular_pwg.hip')], check=True, capture_output=True)
        lines = open(asm).read().split('\n')
    on, body = False, []
    for l in lines:
        if l.startswith('_ZN12_GLOBAL__N_19k_tab_pwgILb0ELb1EEEvNS_8pwg_argsE:'):
            on = True
        elif on and l.startswith('.Lfunc_end'):
            break
        elif on:
            body.append(l)
    # depth-2 loops: from a "Loop Header: Depth=2" label to the next depth-2 header / function end
    heads = [i for i, l in enumerate(body) if 'Loop Header: Depth=2' in l]
    for k, h in enumerate(heads):
        e = heads[k + 1] if k + 1 < len(heads) else len(body)
        seg = [l.strip() for l in body[h:e] if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        if len(seg) < 150:
            continue
        c = {'v_': 0, 's_': 0, 'ds_': 0, 'global_': 0}
        for t in seg:
            for p in c:
                if t.startswith(p):
                    c[p] += 1
        br = sum(1 for t in seg if t.startswith(('s_cbranch', 's_branch')))
        ex = sum(1 for t in seg if 'exec' in t and t.startswith('s_'))
        print('loop at +%d: %d instructions: VALU %d, SALU %d (branches %d, exec-mask ops %d), LDS %d, VMEM %d'
              % (h, len(seg), c['v_'], c['s_'], br, ex, c['ds_'], c['global_']))

if __name__ == '__main__':
    main()
