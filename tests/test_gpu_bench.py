"""bench.py end to end on one GPU: the JSON contract, and the N > 1 path rehearsed with two ranks on
the same card (gloo for the one collective; RCCL needs one GPU per rank) — splitting the instances
over ranks changes nothing in what the monitors report."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--also', '', '--min-seconds', '0']


def _line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('config,instances', [('C3', 768), ('C4', 96)])
def test_two_ranks_split_the_instances_and_report_the_same_monitors(config, instances):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    args = ['--config', config, '--instances', str(instances), '--env-steps', '64'] + COMMON
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args,
                         capture_output=True, text=True, timeout=900, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                          '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                          str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend',
                          'gloo'] + args, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = _line(one.stdout), _line(two.stdout)
    for r, n in ((a, 1), (b, 2)):
        assert r['n_gpus'] == n and r['steps'] == 2 and r['warmup'] == 1
        assert r['scaling'] == 'strong' and r['unit'] == 'env-steps/s' and r['value'] > 0
        assert r['config']['instances_total'] == instances
        assert r['config']['instances_per_gpu'] == instances // n
        assert r['roofline']['frac'] <= 1.0 and r['roofline']['limiter'] in ('issue', 'hbm', 'latency')
    assert b['monitors']['collectives_in_timed_region'] == 1
    for key in ('trials_finished', 'escape_latency_sum', 'trial_reward_sum'):
        assert a['monitors'][key] == b['monitors'][key], key
    assert a['monitors']['trials_finished'] > 0


def test_weak_scaling_option_runs_the_full_count_per_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                          '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                          str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                          '--weak', '--config', 'C2', '--instances', '256', '--env-steps', '50']
                         + COMMON[:-1] + ['0.2'], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out.stdout)
    assert r['scaling'] == 'weak' and r['config']['instances_total'] == 512
    assert r['config']['instances_per_gpu'] == 256
    # --min-seconds: the same two-launch window again and again (both ranks agree when to stop);
    # the headline stays the first window
    rep = r['repeat_windows']
    assert rep['count'] >= 1 and rep['seconds'] >= 0.2 and r['steps'] == 2
    assert rep['ms_per_step_min'] <= rep['ms_per_step_median'] <= rep['ms_per_step_max']
