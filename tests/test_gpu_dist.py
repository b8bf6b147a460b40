"""GPU: the data-parallel mode with TWO ranks (both on the one GPU of the test box, gloo rendezvous on 127.0.0.1).
With synchronised BatchNorm the 2-rank step must reproduce the reference's single-process step on the 2-scene batch
(golden fixture); with per-rank statistics it must not (different normalisation)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world2(sync):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, '_sync_bn_worker.py'), str(sync)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
    return outs


def test_two_ranks_sync_bn_reproduce_single_process_reference(hip):
    outs = _run_world2(1)
    assert all('sync=1' in o for o in outs)


def test_two_ranks_local_bn_is_a_different_model(hip):
    _run_world2(0)


def test_rccl_collectives_single_rank_smoke(hip):
    """every collective of the data-parallel mode on the RCCL backend (one rank: the test box has one GPU)"""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), 'tools', 'rccl_smoke.py')], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and 'rccl smoke ok' in out.stdout, (out.stdout + out.stderr)[-3000:]
