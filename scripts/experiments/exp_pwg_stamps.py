"""Where a ticket's time goes in k_tab_pwg (a library built with -DCOBEL_PWG_STAMPS, COBEL_LIB=...):
cycles per phase — ticket draw, prologue, steps, write-back — summed per wave by the kernel into the
scratch area, read back here and split by the two kinds of wave.

    COBEL_LIB=$PWD/cobel-rl_amd/lib/libcobel_S.so python scripts/experiments/exp_pwg_stamps.py [instances] [pretrain launches]
"""
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cobel-rl_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pre = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    dev = torch.device('cuda', 0)
    cfg = dict(bench.CONFIGS['C3'])
    env, agent = bench.build_agent('C3', cfg, n, 0, dev)
    run = bench.Runner(cfg, env, agent)
    for _ in range(pre):
        run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run.launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    what = run.describe()
    waves = what['instances_per_workgroup']
    sc = agent._scratch.cpu().numpy().view(np.uint32)
    grid = 256
    slices = len(os.environ.get('COBEL_DEBUG_PWG_SLICES', '512').split(','))
    off = 256 + 8 * ((n + 7) // 8) * (slices - 1)
    st = sc[off:off + grid * waves * 8].reshape(grid, waves, 8).astype(np.float64)
    nl = waves
    print('launch %.3f ms, %d waves per workgroup' % (ms, waves))
    for name, sel in (('LDS waves', st[:, :nl]), ('global-memory waves', st[:, nl:])):
        if sel.size == 0:
            continue
        cnt = sel[..., 4].sum()
        tot = sel[..., :4].sum()
        print('%s: %.1f tickets per wave; cycles per ticket: draw %.0f, prologue %.0f, steps %.0f, '
              'write-back %.0f; share outside the steps %.2f %%; total cycles per wave %.3g'
              % (name, cnt / sel[..., 4].size, sel[..., 0].sum() / cnt, sel[..., 1].sum() / cnt,
                 sel[..., 2].sum() / cnt, sel[..., 3].sum() / cnt,
                 100 * (1 - sel[..., 2].sum() / tot), tot / sel[..., 4].size))


    # by wave slot of the workgroup: the SIMD's arbiter serves its oldest wave first (wave w runs on SIMD w % 4)
    with np.errstate(all='ignore'):
        per = st[..., 2].sum(axis=0) / st[..., 4].sum(axis=0)
    print('cycles in the steps per ticket, by wave slot: ' + '  '.join('%d: %.0f' % (w, per[w]) for w in range(waves)))
    ss = sc[off + 26624:off + 26624 + 3].astype(np.float64)
    if ss.any():
        tickets = st[..., 4].sum()
        print('cycles per step, all waves: steps 0-3 %.0f, 4-15 %.0f, 16-63 %.0f, later (LDS waves) %.0f'
              % (ss[0] / tickets / 4, ss[1] / tickets / 12, ss[2] / tickets / 48,
                 st[:, :nl, 2].sum() / st[:, :nl, 4].sum() / max(1, (512 // slices) - 64)))
    # when each XCD's waves ran out of tickets (100 MHz clock), relative to the first start
    raw = sc[off:off + grid * waves * 8].reshape(grid, waves, 8)
    t0 = raw[..., 7].min()
    end = (raw[..., 6] - t0).astype(np.int64) / 100.0          # us
    xcc = raw[..., 5]
    for kind, sl in (('LDS', slice(0, nl)), ('global', slice(nl, waves))):
        if sl.start >= waves:
            continue
        e = end[:, sl].ravel()
        print(kind, 'waves finish (us): percentiles 1 / 10 / 50 / 90 / 99 / max  %s;  idle before the last '
              'wave ends: %.1f %% of the slot time' % ('  '.join('%.0f' % v for v in np.percentile(
                  e, [1, 10, 50, 90, 99, 100])), 100 * (1 - e.mean() / end.max())))
        print(kind, 'waves finish (us) by XCD: ' + '  '.join(
            '%d: %.0f-%.0f (median %.0f)' % (x, end[:, sl][xcc[:, sl] == x].min(),
                                           end[:, sl][xcc[:, sl] == x].max(),
                                           np.median(end[:, sl][xcc[:, sl] == x]))
            for x in range(8) if (xcc[:, sl] == x).any()))


if __name__ == '__main__':
    main()
