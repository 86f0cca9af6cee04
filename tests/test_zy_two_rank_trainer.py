"""GPU: two ranks of trainer.MinibatchTrainer on the one GPU (gloo between them, started by
torch.distributed.run as child processes) against one rank, end to end through tools/two_rank_trainer.py:
the ranks agree bit for bit after 60 iterations including a densification, and with the one-rank run to
rounding after ten."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_ranks_train_like_one(tmp_path):
    tool = os.path.join(ROOT, "tools", "two_rank_trainer.py")
    single, ranks = str(tmp_path / "single.npz"), str(tmp_path / "rank")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, tool, "--out", single], capture_output=True, text=True, timeout=150, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    port = 29700 + os.getpid() % 200
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), tool, "--out", ranks],
                         capture_output=True, text=True, timeout=150, cwd=ROOT, env=env)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-1500:])
    cmp_ = subprocess.run([sys.executable, tool, "--compare", single, ranks + "0.npz", ranks + "1.npz"], capture_output=True,
                          text=True, timeout=100, cwd=ROOT, env=env)
    assert cmp_.returncode == 0, (cmp_.stdout, cmp_.stderr[-1500:])
    assert "agree bit for bit" in cmp_.stdout and "after 10 iterations" in cmp_.stdout
