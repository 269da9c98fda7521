"""bench.py's launch contract (driver: `python bench.py --gpus N ...`; also `python -m torch.distributed.run ... bench.py --gpus N`).

`--dry-run` keeps everything of the N-rank path except the kernels: self-launch of one rank per GPU, 127.0.0.1 rendezvous,
barrier-bracketed timing with MAX over ranks, exactly one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _json_lines(out):
    rows = []
    for line in out.splitlines():
        line = line.strip()
        if line.startswith('{') and line.endswith('}'):
            rows.append(json.loads(line))
    return rows


def test_driver_form_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run', '--steps', '4', '--warmup', '1'], capture_output=True,
                       text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = _json_lines(r.stdout)
    assert len(rows) == 1, r.stdout
    row = rows[0]
    assert row['n_gpus'] == 2 and row['steps'] == 4 and row['warmup'] == 1 and row['config']['parallelism'] == 'dp2'
    # rank 1 sleeps twice as long per step as rank 0: the reported time must be the slower rank's
    assert row['ms_per_step'] >= 4.0 * 0.95


def test_env_launch_form_still_works():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), BENCH, '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '0']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = _json_lines(r.stdout)
    assert len(rows) == 1 and rows[0]['n_gpus'] == 2


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], capture_output=True, text=True, timeout=300, cwd=REPO, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)
