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
COMMON = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--also', '', '--no-pretrain',
          '--min-seconds', '0']


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
        assert r['roofline']['bound'] == 'hbm'
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


def test_plain_invocation_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (as a driver may call it): bench.py
    starts the two ranks itself, as fresh processes, and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    args = ['--config', 'C3', '--instances', '768', '--env-steps', '64'] + COMMON
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args,
                         capture_output=True, text=True, timeout=900, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend',
                          'gloo'] + args, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert b['n_gpus'] == 2 and b['config']['instances_per_gpu'] == 384
    assert b['monitors']['collectives_in_timed_region'] == 1
    for key in ('trials_finished', 'escape_latency_sum', 'trial_reward_sum'):
        assert a['monitors'][key] == b['monitors'][key], key


def test_five_ranks_on_one_card_with_an_uneven_split():
    """More ranks than two, and a total that does not divide (1 003 instances over five ranks: 201,
    201, 201, 200, 200), on ONE card over gloo — five is what a one-GPU box admits next to this
    process; the eight-rank split is rehearsed without a card in tests/test_host_cpu.py."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    args = ['--config', 'C3', '--instances', '1003', '--env-steps', '64'] + COMMON
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args,
                         capture_output=True, text=True, timeout=900, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    five = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '5', '--backend',
                           'gloo'] + args, capture_output=True, text=True, timeout=900, env=env)
    assert five.returncode == 0, five.stderr[-3000:]
    a, b = _line(one.stdout), _line(five.stdout)
    assert b['n_gpus'] == 5 and b['config']['instances_total'] == 1003
    assert b['config']['instances_per_gpu'] == 201          # (rank 0's shard)
    assert b['monitors']['collectives_in_timed_region'] == 1
    for key in ('trials_finished', 'escape_latency_sum', 'trial_reward_sum'):
        assert a['monitors'][key] == b['monitors'][key], key


def test_two_gpus_over_rccl_report_the_same_monitors():
    """The real collective: one rank per GPU, backend "nccl" (= RCCL).  Needs two cards."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('one GPU: RCCL needs one card per rank (the gloo tests rehearse the path)')
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    args = ['--config', 'C3', '--instances', '768', '--env-steps', '64'] + COMMON
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args,
                         capture_output=True, text=True, timeout=900, env=env)
    two = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + args,
                         capture_output=True, text=True, timeout=900, env=env)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr[-2000:], two.stderr[-2000:])
    a, b = _line(one.stdout), _line(two.stdout)
    assert b['n_gpus'] == 2
    assert a['monitors'] == dict(b['monitors'], collectives_in_timed_region=0)


def test_every_default_leg_runs_and_carries_a_roofline(tmp_path):
    """The default command at 1/50 of the instance counts: no leg may end in an "error" key (round
    2's C6 leg did, unnoticed) and every leg states its roofline.  The stdout line is the COMPACT
    object (round 4's full line of 22 KB was not taken by the driver's parser): under 6 000 bytes,
    every key of the contract, a dtype on every leg; the full objects are in --full-out."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    full_path = str(tmp_path / 'bench_full.json')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--scale', '0.02',
                          '--steps', '2', '--warmup', '1', '--min-seconds', '0', '--max-pretrain',
                          '24', '--full-out', full_path], capture_output=True, text=True, timeout=1500,
                         env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(line) == 1 and out.stdout.strip() == line[0], out.stdout[-2000:]
    assert len(line[0]) < 6000, len(line[0])
    c = json.loads(line[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline',
                'pretraining', 'young_agents', 'other_configs'):
        assert key in c, key
    for key in ('bound', 'limiter', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel',
                'algorithmic_bytes_per_env_step', 'issue'):
        assert key in c['roofline'], key
    assert c['roofline']['issue']['valu_per_step'] > 0
    assert c['cpu_baseline']['value'] > 0 and c['cpu_baseline']['cores'] == 1
    assert c['cpu_baseline']['kind'] == 'port' and c['cpu_baseline']['unit'] == 'env-steps/s'
    assert 'workload' in c['config'] and 'model' not in c['config']
    for name, leg in c['other_configs'].items():
        assert set(leg) >= {'value', 'unit', 'dtype', 'bound', 'frac'}, (name, leg)
        assert leg['dtype'] in ('f32', 'f64'), (name, leg)
    r = json.load(open(full_path))
    assert c['value'] == pytest.approx(r['value'], rel=1e-5)
    assert c['roofline']['frac'] == pytest.approx(r['roofline']['frac'], rel=1e-5)
    assert set(c['other_configs']) == set(r['other_configs'])
    assert r['metric'].startswith('gridworld env-steps/sec') and r['n_gpus'] == 1
    assert r['cpu_baseline']['value'] > 0 and r['cpu_baseline']['cores'] == 1
    roof = r['roofline']
    assert 0 < roof['frac'] <= 1.0 and roof['planning_batches_evaluated'] <= roof['planning_batches_drawn']
    # the bytes follow the work the kernel counted, not the work drawn
    per_launch = r['config']['instances_per_gpu'] * r['config']['env_steps_per_launch']
    want = 78 * per_launch + 31 * 50 * roof['planning_batches_evaluated'] // r['steps']
    assert roof['algorithmic_bytes_per_launch'] == want
    # `warmup` echoes the request; the untimed pre-training is counted apart from it
    assert r['warmup'] == 1 and 'young_agents' in r
    assert r['pretraining']['untimed_launches_in_all'] == r['pretraining']['launches'] + 1
    legs = r['other_configs']
    assert set(legs) >= {'C1', 'C2', 'C4', 'C6', 'C5_f64', 'C5_f32', 'dyna_dqn', 'dyna_dsr', 'grid_search',
                         'general_hex_q', 'general_wide_q', 'general_wide_q_lane', 'general_dynaq_b100'}
    for name, leg in legs.items():
        assert 'error' not in leg, (name, leg)
        assert leg['roofline'] is not None, name
        if name != 'C1':     # (one instance = one wavefront: the plumbing case carries no fraction)
            assert leg['roofline']['frac'] > 0, name
        assert leg['value'] > 0, name
    # the LDS-resident legs are held against instruction issue, not against HBM bytes
    for name in ('C2', 'C6', 'grid_search'):
        assert legs[name]['roofline']['bound'] == 'issue', name
        assert legs[name]['roofline']['hbm_accounting']['frac'] > 0, name
    assert roof['bound'] == 'hbm' and roof['issue']['valu_per_step'] > 0
    # the reference loop's port is timed beside every leg that has one
    for name in ('C1', 'C2', 'C4', 'C6'):
        cpu = legs[name]['cpu_baseline']
        assert cpu['value'] > 0 and cpu['cores'] == 1 and cpu['kind'] == 'port', name
        assert 1.0 <= cpu['port_over_reference'][0] <= cpu['port_over_reference'][1], name


def test_one_rank_process_group_over_rccl():
    """What a one-GPU box can execute of the multi-GPU path: a process group of ONE rank on the
    "nccl" backend (= RCCL) — barriers, the MIN all-reduce that ends the pre-training and the monitor
    all-gather are the same calls, tensor types and unpacking as with eight ranks."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    args = ['--config', 'C3', '--instances', '1024', '--env-steps', '64', '--steps', '2', '--warmup', '1',
            '--no-cpu-baseline', '--also', '', '--min-seconds', '0', '--max-pretrain', '17']
    plain = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args,
                           capture_output=True, text=True, timeout=900, env=env)
    assert plain.returncode == 0, plain.stderr[-3000:]
    rccl = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dist-single', '--backend',
                           'nccl'] + args, capture_output=True, text=True, timeout=900, env=env)
    assert rccl.returncode == 0, rccl.stderr[-3000:]
    a, b = _line(plain.stdout), _line(rccl.stdout)
    # (RCCL prints a version banner on file descriptor 1 at communicator creation: bench.py keeps
    #  its stdout to the one JSON line all the same)
    assert rccl.stdout.strip().count('\n') == 0 and plain.stdout.strip().count('\n') == 0
    assert a['monitors']['collectives_in_timed_region'] == 0
    assert b['monitors']['collectives_in_timed_region'] == 1 and b['n_gpus'] == 1
    assert a['warmup'] == b['warmup'] == 1
    assert a['pretraining']['untimed_launches_in_all'] == b['pretraining']['untimed_launches_in_all'] == 17
    for key in ('trials_finished', 'escape_latency_sum', 'trial_reward_sum'):
        assert a['monitors'][key] == b['monitors'][key], key
