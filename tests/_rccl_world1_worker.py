"""Child process of tests/test_gpu_dist.py: a world-size-1 `nccl` (= RCCL) process group on the GPU box, through which the
product's multi-GPU code runs for real -- ShardedPredictor.predict (all-gather of theta) and val_step with
distributed.reduce_losses (the single all-reduce of the [n_stage,4] loss block) -- and must reproduce the ungrouped
single-process results bit for bit.  HPE_FORCE_DIST=1 makes the collectives run even though the world has one rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HPE_FORCE_DIST="1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpe_amd  # noqa: E402
from hpe_amd import distributed as D, synthetic  # noqa: E402


def main():
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = torch.distributed.all_reduce, torch.distributed.all_gather_into_tensor

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return real_ag(*a, **k)

    torch.distributed.all_reduce, torch.distributed.all_gather_into_tensor = count_ar, count_ag

    class Cfg(object):
        img_size, num_stage, batch_size, data_format = 224, 3, 6, "NHWC"
        checkpoint_dir = smpl_model_path = None

    p = hpe_amd.Predictor(Cfg(), smpl_model=synthetic.make_smpl_model(), mean_params=synthetic.make_mean_params(),
                          encoder_params=synthetic.make_encoder_params(), regressor_params=synthetic.make_regressor_params())
    sp = D.ShardedPredictor(p)
    assert (sp.rank, sp.world) == (0, 1)
    img = torch.from_numpy(synthetic.make_images(6, seed=444)).cuda()
    seg, kp_gt = synthetic.make_lsp_targets(6, seed=445)
    # ---- predict: theta_all comes back from RCCL's all-gather and must equal the local theta
    plain = p.predict(img)
    res = sp.predict(img)
    assert calls["all_gather"] == 1, calls
    assert torch.equal(res["theta_all"], plain["theta"]) and res["theta_all"].data_ptr() != res["theta"].data_ptr()
    for k in ("generated_verts", "generated_joints", "generated_cams", "theta"):
        assert torch.equal(res[k], plain[k]), k
    # ---- val_step: ONE all-reduce for the losses of all stages
    calls["all_reduce"] = 0
    ungrouped = p.val_step(img, seg, kp_gt)
    assert calls["all_reduce"] == 0
    grouped = sp.val_step(img, seg, kp_gt)
    assert calls["all_reduce"] == 1, calls
    for a, b in zip(grouped["kpr_losses"] + grouped["mr_losses"], ungrouped["kpr_losses"] + ungrouped["mr_losses"]):
        assert float(a) == float(b), (float(a), float(b))
    assert torch.equal(grouped["loss_parts"][:, [0, 1, 3]], ungrouped["loss_parts"][:, [0, 1, 3]])
    torch.cuda.synchronize()
    torch.distributed.destroy_process_group()
    print("RCCL_WORLD1_OK kpr=%s mr=%s" % ([round(float(x), 5) for x in grouped["kpr_losses"]], [round(float(x), 5) for x in grouped["mr_losses"]]))


if __name__ == "__main__":
    main()
