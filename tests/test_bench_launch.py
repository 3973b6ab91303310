"""bench.py --gpus N must run N ranks when the driver starts it WITHOUT a launcher: it spawns them itself
under torch.distributed.run; a rank count that differs from --gpus is an error, never a silent 1-GPU run.  CPU only (gloo,
--dry-run: process group + one all-gather, no GPU call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(kw)
    return env


def test_gpus2_without_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--dry-run'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks'] == [0, 1] and line['backend'] == 'gloo'


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--dry-run'], env=_env(WORLD_SIZE='1', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'exactly one rank per GPU' in (r.stderr + r.stdout)


def test_single_gpu_dry_run_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, '--dry-run'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1
