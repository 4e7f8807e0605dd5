"""Multi-GPU path on the one-GPU box (-m gpu): the RCCL process group and bench.py's own rank launcher, each in a fresh
child process (the test process itself may already hold a HIP context; it never exec()s, it starts children)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HPE_FORCE_DIST"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(kw)
    return env


def test_sharded_predictor_and_loss_reduce_over_rccl_world1():
    """ShardedPredictor.predict + val_step(reduce_fn=distributed.reduce_losses) over a world-size-1 nccl group: bit-equal
    theta, equal losses, exactly one all-gather per predict and one all-reduce per val_step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_world1_worker.py"), str(_free_port())], env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_WORLD1_OK" in r.stdout


@pytest.mark.parametrize("extra", [[], ["--config5"]])
def test_bench_launches_its_own_ranks(extra):
    """`HPE_FORCE_DIST=1 python bench.py --gpus 1`: the plain command starts its rank process itself (the parent never
    touches HIP), the rank builds an RCCL group, runs the all-gather (and with --config5 the one all-reduce) and exactly one
    JSON line comes back on stdout."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--cpu-sample", "0", "--sustain", "0"] + extra
    r = subprocess.run(cmd, env=_env(HPE_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "RCCL all-gather" in d["config"]["workload"]
    # the N > 1 line verifies itself: every rank checks its slice of the gathered theta against its own theta and two images of
    # its last batch against the oracle; the flags are all-reduced (MIN) into the one JSON line
    dc = d["dist_check"]
    assert dc["ranks"] == 1 and dc["gather_slice_equals_local_theta_on_every_rank"] is True and dc["parity_pass_on_every_rank"] is True
    assert 0 < dc["worst_gated_over_ranks"] < 1e-4 and dc["images_checked_per_rank"] == 2
    if extra:
        assert len(d["losses_last_step"]["kpr"]) == 3 and d["loss_roofline"]["achieved"] > 0
        assert d["loss_roofline"]["frac"] <= 1.0 and d["loss_roofline"]["mfma_grid_search"] + d["loss_roofline"]["mfma_full_search"] > 0
