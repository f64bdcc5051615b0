"""The RCCL path of cremage_amd.dist on real hardware at world size 1: a fresh child process whose first GPU call is
`init_process_group("nccl", world_size=1, device_id=...)`, then the parameter broadcast and the batch all-gather on device tensors.
The world-size-2 semantics are covered on CPU by tests/test_dist_cpu.py (gloo); an 8-GPU node is the driver's to run."""
import os
import subprocess
import sys

import pytest

from tests.conftest import REPO

pytestmark = pytest.mark.gpu


def test_rccl_world1_broadcast_and_all_gather():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_rccl_ws1_run.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "RCCL_WS1_OK" in r.stdout
