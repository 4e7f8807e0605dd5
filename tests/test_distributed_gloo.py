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


# ------------------------------------------------------------------------------------------- world-8 rehearsal (round 4)
def _load_bench():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _theta_rows(lo, hi):
    """deterministic stand-in for a rank's theta rows (the collective plumbing is under test, not the arithmetic)"""
    i = np.arange(lo, hi, dtype=np.float64)[:, None]
    k = np.arange(85, dtype=np.float64)[None, :]
    return np.sin(0.37 * i + 0.11 * k).astype(np.float32)


def _worker8(rank, world, port, n_global, failing_rank, out_dir, shm_tag):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TORCHELASTIC_RUN_ID=shm_tag)
    os.environ.pop("HPE_BENCH_ASSETS", None)
    torch.set_num_threads(1)
    from hpe_amd import distributed as D

    bench = _load_bench()
    # (1) the asset hand-over under a torchrun-style launch: local rank 0 writes the node's one file into /dev/shm, seven ranks wait for it
    tiny = dict(reg={"dense_0/kernel": np.arange(12, dtype=np.float32).reshape(3, 4)}, smpl={"kintree_table": np.arange(48, dtype=np.uint32).reshape(2, 24)})
    bench.make_assets = lambda: tiny
    if rank != 0:
        import time

        time.sleep(0.05 * rank)  # stagger the waiters
    assets, made = bench.get_assets(world, rank)
    assert (made is not None) == (rank == 0)
    assert np.array_equal(assets["reg"]["dense_0/kernel"], tiny["reg"]["dense_0/kernel"]) and assets["smpl"]["kintree_table"].dtype == np.uint32

    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # (2) shard bounds + the one all-gather of theta (equal shards at 2,048 images; ragged when n_global % world != 0)
    lo, hi = D.shard_bounds(n_global, rank, world)
    counts = [D.shard_bounds(n_global, q, world)[1] - D.shard_bounds(n_global, q, world)[0] for q in range(world)]
    t_local = torch.from_numpy(_theta_rows(lo, hi))
    theta_all = D.all_gather_theta(t_local) if len(set(counts)) == 1 else D.all_gather_theta_ragged(t_local, counts)
    slice_ok = bool(torch.equal(theta_all[lo:hi], t_local))
    # (3) the config-5 loss block: exactly one all-reduce of [3, 4]
    packed_local = torch.tensor([[float(rank + 1) * (s + 1), 2.0 * (hi - lo), -1.0, float(rank) + 0.5 * s] for s in range(3)], dtype=torch.float64)
    n_calls = [0]
    real = torch.distributed.all_reduce

    def counting(*a, **k):
        n_calls[0] += 1
        return real(*a, **k)

    torch.distributed.all_reduce = counting
    packed = D.reduce_losses(packed_local)
    torch.distributed.all_reduce = real
    assert n_calls[0] == 1
    # (4) dist_check: one rank's parity is forced to fail -> the MIN-reduced verdict fails on EVERY rank (bench.py then exits 3 everywhere)
    chk = bench.reduce_dist_check(torch, torch.distributed, world, slice_ok, rank != failing_rank, 1e-6 * (rank + 1) if rank != failing_rank else 0.5, "cpu")
    np.save(os.path.join(out_dir, "verdict_%d.npy" % rank), np.array([float(bench.dist_check_failed(chk)), chk["worst_gated_over_ranks"],
                                                                      float(chk["gather_slice_equals_local_theta_on_every_rank"])]))
    if rank == 0:
        np.save(os.path.join(out_dir, "theta_all.npy"), theta_all.numpy())
        np.save(os.path.join(out_dir, "packed.npy"), packed.numpy())
    torch.distributed.barrier()
    if made and os.path.exists(made):
        os.remove(made)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_global,failing_rank", [(2048, -1), (2043, 5)])
def test_world8_rehearsal(tmp_path, n_global, failing_rank):
    """What the first 8-GPU run of bench.py does around the device work, with 8 gloo ranks on CPU: asset hand-over through /dev/shm
    (one writer, seven waiters), shard bounds at 8 x 256 images, equal and ragged theta gather, the single [3, 4] loss all-reduce, and
    the self-check verdict with one rank forced to fail (every rank must see the failure: that is what makes all of them exit 3)."""
    world = 8
    port = _free_port()
    tag = "pytest%d_%d" % (os.getpid(), n_global)
    mp.spawn(_worker8, args=(world, port, n_global, failing_rank, str(tmp_path), tag), nprocs=world, join=True)
    theta_all = np.load(tmp_path / "theta_all.npy")
    assert theta_all.shape == (n_global, 85)
    np.testing.assert_array_equal(theta_all, _theta_rows(0, n_global))
    packed = np.load(tmp_path / "packed.npy")
    for s in range(3):
        num, cnt = 36.0 * (s + 1), 2.0 * n_global  # sum of (rank + 1) over 8 ranks = 36
        assert packed[s, 0] == num and packed[s, 1] == cnt and abs(packed[s, 2] - num / cnt) < 1e-12
        assert packed[s, 3] == 28.0 + 8 * 0.5 * s
    for r in range(world):
        failed, worst, gather_ok = np.load(tmp_path / ("verdict_%d.npy" % r))
        assert gather_ok == 1.0
        assert bool(failed) == (failing_rank >= 0), (r, failed)
        assert abs(worst - (0.5 if failing_rank >= 0 else 8e-6)) < 1e-12
    assert not [f for f in os.listdir("/dev/shm") if tag in f], "the node's asset file was not removed"


def test_cpu_split_of_a_two_node_host_into_8_disjoint_sets():
    """pin_rank_to_gpu_numa's plan: a 128-core, two-NUMA-node host (sysfs cpulists with hyperthread ranges), GPUs 0-3 on node 0 and 4-7
    on node 1 -> eight disjoint sets of 16 cores, each inside its GPU's node; unknown nodes and uneven splits degrade gracefully."""
    bench = _load_bench()
    node_cpus = {0: bench.parse_cpulist("0-31,64-95\n"), 1: bench.parse_cpulist("32-63,96-127\n")}
    node_of_rank = [0, 0, 0, 0, 1, 1, 1, 1]
    sets = [bench.plan_rank_cpus(node_of_rank, node_cpus, r) for r in range(8)]
    assert all(len(s) == 16 for s in sets)
    assert sorted(c for s in sets for c in s) == list(range(128))  # disjoint and complete
    for r, s in enumerate(sets):
        assert set(s) <= set(node_cpus[node_of_rank[r]])
    # 8 GPUs on ONE node of 20 usable cores: 3,3,3,3,2,2,2,2
    sets = [bench.plan_rank_cpus([0] * 8, {0: list(range(20))}, r) for r in range(8)]
    assert [len(s) for s in sets] == [3, 3, 3, 3, 2, 2, 2, 2] and sorted(c for s in sets for c in s) == list(range(20))
    assert bench.plan_rank_cpus([-1, 0], {0: [0, 1]}, 0) is None       # numa_node = -1: no pinning
    assert bench.plan_rank_cpus([0] * 8, {0: [0, 1, 2]}, 7) is None    # fewer cores than ranks: the late ranks stay unpinned
    assert bench.cpu_share(list(range(10)), 3, 0) == [0, 1, 2, 3] and bench.cpu_share(list(range(10)), 3, 2) == [7, 8, 9]
