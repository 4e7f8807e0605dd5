#!/usr/bin/env python3
"""bench.py -- images/sec of the per-image forward hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole path over one batch of 256 synthetic 224x224x3 images per GPU, already
resident in HBM: ResNet-50 v1 encoder (fp32 MFMA) -> 3 IEF regressor stages -> SMPL (LBS, joint regress,
orthographic reprojection) at ALL three stages (what Trainer.val_step evaluates; nothing is skipped) ->
for N > 1 one RCCL all-gather of the final theta [256,85] per rank over xGMI.  Images shard by batch
(independent units, no data-path collective besides that gather): weak scaling.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family, the 53 convolution layers (implicit-GEMM
conv_gemm_f32_dma_kernel; the 3x3 layers run as fp32 Winograd F(2x2,3x3), which does 2.25x fewer multiplies for
the same layer -- the algorithmic FLOPs priced here are the direct convolution's): achieved = 7.7119 GFLOP/img * 256 img / (encoder span of the last
timed step, HIP events recorded on the launch stream; the batch-chunk streams overlap their conv launches, so the
span -- not a sum of overlapping durations -- is the family's time), peak = 157.3 TFLOP/s fp32 MFMA.
`roofline.serial` is the same quantity with the chunk streams off and events around each launch (one extra step
after the timed region); it is the number the rocprofv3 kernel stats in profiles/ add up to.
`cpu_baseline` is the CPU oracle (a NumPy / torch-CPU restatement of the reference path -- TensorFlow is not
installable here) timed on this host's cores on a bounded sample, rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

# The encoder runs its batch chunks on 3 HIP streams and RCCL adds streams of its own; the HIP runtime multiplexes all
# streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and once two chunk streams share a queue
# their overlap is lost (measured: 20.1 -> 18.4 ms/step with a process group alive).  Must be set before HIP initialises.
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENCODER_GFLOP_PER_IMG = 7.711850496  # 2 * 3,855,925,248 MAC (resnet_spec.encoder_macs_per_image)
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same guide)


def main():
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # see the print at the end
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE metric: 256)")
    ap.add_argument("--cpu-sample", type=int, default=96, help="images of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--encoder-dtype", default="fp32", choices=["fp32", "bf16"], help="bf16 = BASELINE configs[3] (bf16 encoder, fp32 SMPL)")
    ap.add_argument("--config5", action="store_true", help="also evaluate kp + mesh reprojection losses of every stage (BASELINE configs[4])")
    args = ap.parse_args()

    import numpy as np
    import torch

    import hpe_amd
    from hpe_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("HPE_FORCE_DIST"))  # HPE_FORCE_DIST: rehearse the RCCL calls at world 1
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = args.batch
    smpl = synthetic.make_smpl_model()
    enc = synthetic.make_encoder_params()
    reg = synthetic.make_regressor_params()
    mean_vals = synthetic.make_mean_params()

    class Cfg(object):
        img_size, num_stage, batch_size, data_format = 224, 3, B, "NHWC"
        checkpoint_dir = smpl_model_path = None
        encoder_dtype = args.encoder_dtype

    pred = hpe_amd.Predictor(Cfg(), smpl_model=smpl, mean_params=mean_vals, encoder_params=enc, regressor_params=reg,
                             device=local_rank)
    eng = pred.engine
    images = torch.from_numpy(synthetic.make_images(B, seed=1000 + rank)).cuda()
    want = eng.DEFAULT_OUTPUTS + (("verts2d",) if args.config5 else ())
    # two output sets used alternately: the all-gather of step k reads theta_k while step k+1 already writes theta_{k+1}
    plans = [eng.make_forward_plan(B, all_stages=True, want=want) for _ in range(2)]
    run, outs = plans[0]
    if args.config5:
        from hpe_amd import distributed as D
        from hpe_amd.ops import kp_reprojection_loss

        seg_np, kp_np = synthetic.make_lsp_targets(B, seed=2000 + rank)
        seg_gts = torch.from_numpy(seg_np[..., 0].copy()).cuda()
        kp_gts = torch.from_numpy(kp_np).cuda()
        losses = {}
    theta_all = [torch.empty((world * B, 85), dtype=torch.float32, device="cuda") for _ in range(2)] if use_dist else None
    pending = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        if use_dist and pending[k] is not None:
            pending[k].wait()  # the gather that last used this output set (two steps ago)
            pending[k] = None
        o = plans[k][0](images)
        if args.config5:
            kp, mr = [], []
            for st in o:
                parts = kp_reprojection_loss(kp_gts, st["kp2d"], return_parts=True)
                mesh = eng.mesh_loss(seg_gts, st["verts2d"])
                k5, m5 = D.reduce_losses(parts, mesh) if use_dist else (parts[2], mesh)
                kp.append(60.0 * k5)
                mr.append(0.001 * m5)
            losses["kpr"], losses["mr"] = kp, mr
        if use_dist:
            # the ONE data-path collective: all-gather of the predicted theta over RCCL, asynchronous so that it overlaps
            # the next batch's encoder (it is waited for before its buffers are reused and before the timed region ends)
            pending[k] = dist.all_gather_into_tensor(theta_all[k], o[-1]["theta"], async_op=True)
        return o

    # one-time initialisation that is not a benchmark step: code-object load / first-launch setup of every kernel and the RCCL
    # communicator (both are lazy); the W warm-up steps and the K timed steps follow
    eng.forward(images[:2], all_stages=True)
    if use_dist:
        dist.all_gather_into_tensor(theta_all[0], outs[-1]["theta"])
    torch.cuda.synchronize()

    def fence():
        for k in range(2):
            if use_dist and pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_roofline:
        eng.enable_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    phase = None
    if not args.no_roofline:
        # (1) live, over the timed region: HIP events on the launch stream around the encoder (fork/join of the
        #     batch-chunk streams).  The conv launches of the chunks overlap, so the kernel family's time is the
        #     encoder span (it also contains pad / max-pool / avg-pool, ~2 % -> the fraction is conservative).
        tm = eng.timings()  # events of the last timed step
        span_ms = tm["encoder_ms"]
        achieved = ENCODER_GFLOP_PER_IMG * B / span_ms  # GFLOP/ms == TFLOP/s
        # (2) serial cross-check, extra steps after the timed region: chunk streams off, events around each of the
        #     53 launches; the sum matches the rocprofv3 --kernel-trace --stats durations (profiles/).
        eng.enable_timing(2)
        step()
        fence()
        ts = eng.timings()
        per_conv = eng.conv_timings()
        serial_tf = ENCODER_GFLOP_PER_IMG * B / ts["conv_ms"]
        PEAK = PEAK_FP32_MFMA_TFLOPS if args.encoder_dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
        roofline = {
            "bound": "mfma",
            "kernel": ("conv_gemm_f32_dma_kernel (37 layers) + wino_fused_kernel / wino_input_kernel + wino_gemm_kernel (the 16 3x3 layers "
                       "as fp32 Winograd F(2x2,3x3))" if args.encoder_dtype == "fp32" else "conv_gemm_bf16_dma_kernel")
                      + " -- the 53 conv layers of one step, priced at their direct-convolution FLOPs; batch chunks on %s concurrent streams" % os.environ.get("HPE_STREAMS", "3"),
            "achieved": round(achieved, 3),
            "peak": PEAK,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK, 4),
            "traffic": None,
            "launch_ms": round(span_ms, 4),
            "flop_per_launch": ENCODER_GFLOP_PER_IMG * B * 1e9,
            "serial": {"sum_of_53_launch_ms": round(ts["conv_ms"], 4), "achieved": round(serial_tf, 3),
                       "frac": round(serial_tf / PEAK, 4)},
        }
        if args.encoder_dtype == "bf16":
            # at 16x the fp32 matrix rate the bf16 encoder is HBM bound: price it in algorithmic bytes
            nbytes = hpe_amd.resnet_spec.encoder_min_bytes_per_image(2) * B
            gbs = nbytes / span_ms / 1e6
            roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                             "bytes_per_launch": nbytes, "mfma_tflops": round(achieved, 1)})
            roofline.pop("flop_per_launch")
            roofline["serial"] = {"sum_of_53_launch_ms": round(ts["conv_ms"], 4)}
        if args.encoder_dtype == "fp32":
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run
            # inside the timed region); scaled by batch, null if no measurement of this build family is committed.
            import glob

            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*final_conv_hbm_traffic.json")))
            if cands:
                tj = json.load(open(cands[-1]))
                roofline["traffic"] = round(tj["total_bytes"] * B / tj["batch"])
                roofline["traffic_source"] = os.path.relpath(cands[-1], ROOT)
        phase = {"encoder_ms": round(tm["encoder_ms"], 3), "regress_smpl_ms": round(tm["regress_smpl_ms"], 3),
                 "step_ms_events": round(tm["total_ms"], 3)}
        eng.enable_timing(0)
        if rank == 0 and os.environ.get("HPE_BENCH_LAYERS"):
            for s, ms in zip(hpe_amd.resnet_spec.CONV_SPECS, per_conv):
                fl = 2.0 * s.kh * s.kw * s.cin * s.cout * s.hout * s.hout * B
                print("%-18s %8.3f ms %7.1f TF" % (s.name, ms, fl / ms / 1e9), file=sys.stderr)

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import hmr_oracle as O  # the checker, timed as the CPU baseline

        n = args.cpu_sample
        img_np = images[:n].cpu().numpy()
        osmpl = O.SMPL(smpl)
        mean = O.load_mean_param(mean_vals)
        O.predict(img_np[:1], enc, reg, osmpl, mean)  # warm-up (thread pools, allocator)
        tc = time.perf_counter()
        ref = O.predict(img_np, enc, reg, osmpl, mean)
        tcpu = time.perf_counter() - tc
        cpu_baseline = {
            "value": round(n / tcpu, 3),
            "unit": "images/sec",
            "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": "%d of the %d bench images, full path (ResNet-50 + 3 IEF stages + SMPL x3), CPU restatement of "
                      "the reference (NumPy + torch-CPU conv2d), not TensorFlow" % (n, B),
        }
        last_outs = plans[(step_no[0] - 1) & 1][1]
        j = last_outs[-1]["joints"][:n].cpu().numpy()
        v = last_outs[-1]["verts"][:n].cpu().numpy()
        def _rel(a, b):
            return float(np.abs(a - b).max() / np.abs(b).max())

        parity = {
            # "MPJPE vs ref" of BASELINE.json: mean Euclidean distance to the oracle's joints, same inputs and weights, on the
            # images of the CPU sample taken out of the full-size batch (so the Winograd / chunked paths are what is checked)
            "mpjpe_vs_oracle": float(np.linalg.norm(j - ref["generated_joints"], axis=-1).mean()),
            "verts_rel_err": _rel(v, ref["generated_verts"]),
            "bar": "1e-4 relative fp32",
        }
        if "J_transformed" in last_outs[-1]:
            j24 = last_outs[-1]["J_transformed"][:n].cpu().numpy()
            parity["mpjpe24_vs_oracle"] = float(np.linalg.norm(j24 - ref["J_transformed"], axis=-1).mean())
        if "theta" in last_outs[-1]:
            parity["theta_rel_err"] = _rel(last_outs[-1]["theta"][:n].cpu().numpy(), ref["theta"])
        if "kp2d" in last_outs[-1]:
            parity["kp2d_rel_err"] = _rel(last_outs[-1]["kp2d"][:n].cpu().numpy(), ref["generated_kp2d"])

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "images/sec (224x224, batch 256/GPU)",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.encoder_dtype == "fp32" else "bf16 (encoder; fp32 accumulate, fp32 regressor+SMPL)",
            "data": "synthetic",
            "config": {
                "workload": "batch=%d/GPU 224x224x3 synthetic images, fp32 ResNet-50 v1 + 3-iter regressor + SMPL LBS at all 3 "
                            "stages%s (BASELINE metric config: batch 256/GPU; %s)" % (B, ", RCCL all-gather of theta" if world > 1 else "", "configs[2]" if world > 1 else "configs[1] at the metric batch"),
                "global_batch": world * B,
                "parallelism": "dp%d (batch shard, replicated weights)" % world,
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if args.config5:
            line["config"]["workload"] += " + kp/mesh reprojection losses of all 3 stages (configs[4])"
            line["losses_last_step"] = {"kpr": [float(x) for x in losses["kpr"]], "mr": [float(x) for x in losses["mr"]]}
        if phase:
            line["phase_ms"] = phase
        if parity:
            line["parity"] = parity
        # stdout carries exactly one line: native libraries (RCCL prints a version banner on fd 1 when its communicator is
        # created) wrote to stderr for the whole run, the JSON goes to the real stdout
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        if rank == 0 and world == 1:
            last = (step_no[0] - 1) & 1
            assert torch.equal(theta_all[last], plans[last][1][-1]["theta"]), "all-gather at world 1 must return the local theta"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
