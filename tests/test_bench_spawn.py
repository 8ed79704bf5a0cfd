"""bench.py --gpus N without a launcher starts its own ranks; a rank that dies must end the job (the others
would sit in their next collective for ever).  Without a GPU every rank fails when it creates its context, so
here the whole job has to come back, non-zero, in seconds."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_spawned_job_ends_when_a_rank_fails():
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')      # no device in any rank, wherever this runs
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    t = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--comm', 'shm', '--workload', 'C2',
                        '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-extra'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=170)
    assert p.returncode != 0
    assert time.time() - t < 160
    assert b'"metric"' not in p.stdout                  # no result line from a job that did not run
