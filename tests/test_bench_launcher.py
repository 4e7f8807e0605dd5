"""bench.py's rank launcher without a GPU: `python bench.py --gpus N` must start N rank processes itself (the parent makes
no HIP call), and a failing rank must make the whole command fail without a JSON line.  Here every rank fails for lack of a
device, which is exactly the failure path; the success path runs on the GPU box (tests/test_gpu_dist.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("n", [2, 8])
def test_launcher_propagates_rank_failure(n):
    """n = 8 is the driver's scaling run: eight rank processes, the node's assets written once into /dev/shm by the parent and removed
    again, the first failing rank's code relayed, no JSON line."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU-box counterpart is tests/test_gpu_dist.py)")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    before = set(f for f in os.listdir("/dev/shm") if f.startswith("hpe_bench_assets"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--batch", "8",
                        "--cpu-sample", "0", "--sustain", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank" in r.stderr and "exited with code" in r.stderr
    assert set(f for f in os.listdir("/dev/shm") if f.startswith("hpe_bench_assets")) == before  # the parent cleaned up


def test_launcher_is_bypassed_under_a_launcher_env():
    """With RANK / WORLD_SIZE already in the environment (torchrun) nothing is spawned: the process is a rank itself and
    rejects a world size that does not match --gpus before any device work."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=4 does not match --gpus 2" in r.stderr


def test_assets_roundtrip_and_cpulist(tmp_path):
    """N > 1: the synthetic weights are generated once per node and handed to the ranks as one .npz in /dev/shm; loading it must
    give back bit-identical arrays with their dtypes (the SMPL kintree is uint32).  parse_cpulist reads sysfs cpulists."""
    import importlib.util

    import numpy as np

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("") == []
    from hpe_amd import synthetic

    a = dict(smpl=synthetic.make_smpl_model(), reg=synthetic.make_regressor_params(), reg_bounded=synthetic.make_regressor_params(variant="bounded"),
             mean=synthetic.make_mean_params())
    path = str(tmp_path / "assets.npz")
    bench.save_assets(a, path)
    b = bench.load_assets(path)
    assert set(b) == set(a)
    for grp in a:
        assert set(a[grp]) == set(b[grp])
        for k in a[grp]:
            x, y = np.asarray(a[grp][k]), b[grp][k]
            assert x.dtype == y.dtype and np.array_equal(x, y), (grp, k)
    # the bounded variant is the survey draw with a smaller last-layer step, nothing else
    assert np.array_equal(a["reg"]["dense_0/kernel"], a["reg_bounded"]["dense_0/kernel"])
    assert np.allclose(a["reg_bounded"]["dense_2/kernel"], 0.25 * a["reg"]["dense_2/kernel"])
