"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): sharding, the single all-gather of theta and the
(numerator, count) all-reduce of the keypoint loss.  The per-rank compute is stood in for by the CPU oracle's
regressor (tests may use the oracle); the collective plumbing under test is the product's
``hpe_amd.distributed`` module, the same code bench.py / ShardedPredictor run over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import hpe_amd
    from hpe_amd import distributed as D, synthetic
    from oracle import hmr_oracle as O

    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    reg = synthetic.make_regressor_params()
    mean = O.load_mean_param(synthetic.make_mean_params())
    g = np.random.Generator(np.random.Philox(77))
    feats = np.abs(g.normal(0, 2, (n_global, 2048))).astype(np.float32)
    lo, hi = D.shard_bounds(n_global, rank, world)
    th = np.tile(mean, (hi - lo, 1))
    for _ in range(3):
        th = th + O.regression_network(np.concatenate([feats[lo:hi], th], 1), reg)
    counts = [D.shard_bounds(n_global, q, world)[1] - D.shard_bounds(n_global, q, world)[0] for q in range(world)]
    t_local = torch.from_numpy(th.astype(np.float32))
    if len(set(counts)) == 1:
        theta_all = D.all_gather_theta(t_local)
    else:
        theta_all = D.all_gather_theta_ragged(t_local, counts)
    # keypoint loss: global normalisation needs (numerator, count) reduced separately
    _, kp_gt = synthetic.make_lsp_targets(n_global, seed=5)
    pred = g.uniform(-1, 1, (n_global, 19, 2)).astype(np.float32)
    gt_l, pr_l = kp_gt[lo:hi].reshape(-1, 3), pred[lo:hi].reshape(-1, 2)
    vis = gt_l[:, 2:3]
    num = float((np.abs(pr_l - gt_l[:, :2]) * vis).sum())
    cnt = float(2 * np.count_nonzero(vis))
    loss = D.reduce_kp_loss(torch.tensor([num, cnt, 0.0], dtype=torch.float64))
    total = D.reduce_sum(torch.tensor([float(hi - lo)]))
    # the config-5 block of all stages: ONE all-reduce of [n_stage, 4] = (kp numerator, kp count, kp loss, mesh sum)
    packed_local = torch.tensor([[num * (s + 1), cnt, -1.0, float(rank + 1) * (s + 1)] for s in range(3)], dtype=torch.float64)
    n_calls = [0]
    real = torch.distributed.all_reduce

    def counting(*a, **k):
        n_calls[0] += 1
        return real(*a, **k)

    torch.distributed.all_reduce = counting
    packed = D.reduce_losses(packed_local)
    torch.distributed.all_reduce = real
    assert n_calls[0] == 1, n_calls
    if rank == 0:
        np.save(os.path.join(out_dir, "packed.npy"), packed.numpy())
        np.save(os.path.join(out_dir, "theta_all.npy"), theta_all.numpy())
        np.save(os.path.join(out_dir, "kp_loss.npy"), np.array([float(loss), float(total)]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_global", [8, 7])
def test_world2_gather_and_loss_reduce(tmp_path, n_global):
    from hpe_amd import synthetic
    from oracle import hmr_oracle as O

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_global, str(tmp_path)), nprocs=2, join=True)
    theta_all = np.load(tmp_path / "theta_all.npy")
    reg = synthetic.make_regressor_params()
    mean = O.load_mean_param(synthetic.make_mean_params())
    g = np.random.Generator(np.random.Philox(77))
    feats = np.abs(g.normal(0, 2, (n_global, 2048))).astype(np.float32)
    th = np.tile(mean, (n_global, 1))
    for _ in range(3):
        th = th + O.regression_network(np.concatenate([feats, th], 1), reg)
    assert theta_all.shape == (n_global, 85)
    np.testing.assert_allclose(theta_all, th.astype(np.float32), rtol=0, atol=2e-6)
    _, kp_gt = synthetic.make_lsp_targets(n_global, seed=5)
    pred = g.uniform(-1, 1, (n_global, 19, 2)).astype(np.float32)
    ref = O.kp_reprojection_loss(kp_gt, pred)
    loss, total = np.load(tmp_path / "kp_loss.npy")
    assert abs(loss - ref) < 1e-6 and total == n_global
    packed = np.load(tmp_path / "packed.npy")
    for s in range(3):
        assert abs(packed[s, 2] - (s + 1) * ref) < 1e-5       # numerator and count reduced separately, divided afterwards
        assert packed[s, 3] == 3.0 * (s + 1)                   # mesh sums of ranks 0 and 1: (1 + 2) * (s + 1)


def test_shard_bounds_cover_everything():
    from hpe_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 256, 2048):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
