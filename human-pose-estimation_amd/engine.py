"""HpeEngine: thin owner of one ``hpe_ctx`` (include/hpe.h) -- asset ingestion from Keras-layout dicts and
torch-tensor plumbing around the C-ABI calls.  PyTorch is used for device memory, streams and (in
``distributed.py``) the RCCL process group only; every numerical stage runs in libhpe_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .resnet_spec import CONV_SPECS

NUM_VERTS = _lib.NUM_VERTS


def _torch():
    import torch

    return torch


def _require_cuda_tensor(t, name, shape_tail=None):
    torch = _torch()
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor on the GPU" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError("%s must have shape [B,%s], got %s" % (name, ",".join(map(str, shape_tail)), tuple(t.shape)))
    return t.contiguous()


class HpeEngine(object):
    def __init__(self, device=0, max_batch=8, num_stage=3, bn_eps=1e-3, encoder_dtype="fp32", **plan_options):
        """plan_options: the HpeConfig plan fields of include/hpe.h (n_streams, dual_gemm, stem_fused, wino_min_c, wino_min_items,
        wino_fused, wino_fused_min_hw, mesh_a2b, wino_f4, wino4_fused, bf16_p8, wino4_ksplit, chain_fuse, halo3); unset = -1 = the library default (environment variable, else built-in).
        They select WHICH kernels run, per context -- two engines with different options can coexist in one process."""
        self.lib = _lib.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.HpeError("no GPU visible: the HIP path is the only path (no CPU fallback)")
        self.device = int(device)
        self.max_batch = int(max_batch)
        self.num_stage = int(num_stage)
        self.num_kp = 19
        if encoder_dtype not in ("fp32", "bf16"):
            raise ValueError("encoder_dtype must be 'fp32' or 'bf16'")
        self.encoder_dtype = encoder_dtype
        cfg = _lib.HpeConfig()
        self.lib.hpe_config_init(C.byref(cfg))
        cfg.device, cfg.max_batch, cfg.num_stage, cfg.bn_eps = self.device, self.max_batch, self.num_stage, float(bn_eps)
        cfg.encoder_dtype = 1 if encoder_dtype == "bf16" else 0
        for k, v in plan_options.items():
            if k not in _lib.PLAN_OPTIONS:
                raise TypeError("unknown plan option %r (known: %s)" % (k, ", ".join(_lib.PLAN_OPTIONS)))
            if k == "mesh_a2b" and isinstance(v, str):
                v = {"grid": 0, "valu": 1, "mfma": 2}[v]
            setattr(cfg, k, int(v))
        self.plan_options = dict(plan_options)
        h = C.c_void_p()
        _lib.check(self.lib.hpe_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self.tdev = torch.device("cuda", self.device)

    # ------------------------------------------------------------------ ingestion
    def load_smpl(self, model, joint_type="cocoplus"):
        """model: dict with the reference pickle's keys (src/tf_smpl/batch_smpl.py:31-81); scipy-sparse
        regressors and chumpy-like objects (``.r``) are densified here exactly as the reference does."""

        def dense(x):
            if hasattr(x, "todense"):
                x = np.asarray(x.todense())
            elif hasattr(x, "r") and not isinstance(x, np.ndarray):
                x = np.asarray(x.r)
            return np.asarray(x)

        if joint_type not in ("cocoplus", "lsp"):
            raise ValueError('BAD!! Unknown joint type: %s, it must be either "cocoplus" or "lsp"' % joint_type)
        kp = dense(model["cocoplus_regressor"])
        if joint_type == "lsp":
            kp = kp[:14]
        parents = np.ascontiguousarray(np.asarray(model["kintree_table"])[0].astype(np.int64).astype(np.int32))
        parents[parents < 0] = -1
        parents[0] = -1  # uint32(-1) -> int32 (batch_smpl.py:65)
        keep = [_lib.f32(dense(model[k])) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")]
        keep.append(_lib.f32(kp))
        if keep[0][0].shape != (NUM_VERTS, 3) or keep[1][0].shape != (NUM_VERTS, 3, 10) or keep[2][0].shape != (NUM_VERTS, 3, 207):
            raise ValueError("SMPL arrays have unexpected shapes")
        if keep[3][0].shape != (24, NUM_VERTS) or keep[4][0].shape != (NUM_VERTS, 24) or keep[5][0].shape[1] != NUM_VERTS:
            raise ValueError("SMPL regressor/weight arrays have unexpected shapes")
        m = _lib.HpeSmplModel(*[k[1] for k in keep], parents.ctypes.data_as(C.c_void_p), int(kp.shape[0]))
        _lib.check(self.lib.hpe_load_smpl(self._h, C.byref(m)))
        self.num_kp = int(kp.shape[0])

    def load_encoder(self, params):
        """params: {'<layer>/kernel' HWIO, '<layer>/bias', '<bn>/gamma|beta|moving_mean|moving_variance'}"""
        for i, s in enumerate(CONV_SPECS):
            arrs = [
                _lib.f32(params[s.name + "/kernel"]),
                _lib.f32(params[s.name + "/bias"]),
                _lib.f32(params[s.bn_name + "/gamma"]),
                _lib.f32(params[s.bn_name + "/beta"]),
                _lib.f32(params[s.bn_name + "/moving_mean"]),
                _lib.f32(params[s.bn_name + "/moving_variance"]),
            ]
            if arrs[0][0].shape != (s.kh, s.kw, s.cin, s.cout):
                raise ValueError("%s/kernel must be HWIO %s, got %s" % (s.name, (s.kh, s.kw, s.cin, s.cout), arrs[0][0].shape))
            _lib.check(self.lib.hpe_load_conv(self._h, i, *[a[1] for a in arrs]))

    def load_regressor(self, params):
        dims = [(2133, 1024), (1024, 1024), (1024, 85)]
        for i, d in enumerate(dims):
            k = _lib.f32(params["dense_%d/kernel" % i])
            b = _lib.f32(params["dense_%d/bias" % i])
            if k[0].shape != d:
                raise ValueError("dense_%d/kernel must be %s" % (i, d))
            _lib.check(self.lib.hpe_load_dense(self._h, i, k[1], b[1]))

    def load_mean_theta(self, mean85):
        m = _lib.f32(np.asarray(mean85).reshape(-1))
        if m[0].shape != (85,):
            raise ValueError("mean theta must have 85 entries")
        _lib.check(self.lib.hpe_load_mean_theta(self._h, m[1]))

    def finalize(self):
        _lib.check(self.lib.hpe_finalize(self._h))
        self._finalized = True

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.hpe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.tdev).cuda_stream)

    def _new(self, *shape):
        return _torch().empty(shape, dtype=_torch().float32, device=self.tdev)

    def _alloc_outputs(self, B, want):
        shapes = {
            "verts": (B, NUM_VERTS, 3),
            "joints": (B, self.num_kp, 3),
            "cams": (B, 3),
            "theta": (B, 85),
            "J_transformed": (B, 24, 3),
            "kp2d": (B, self.num_kp, 2),
            "verts2d": (B, NUM_VERTS, 2),
            "Rs": (B, 24, 3, 3),
        }
        tensors = {k: self._new(*shapes[k]) for k in want}
        o = _lib.HpeOutputs(*[tensors[k].data_ptr() if k in tensors else None for k in _lib.OUTPUT_FIELDS])
        return tensors, o

    # ------------------------------------------------------------------ compute
    DEFAULT_OUTPUTS = ("verts", "joints", "cams", "theta", "J_transformed", "kp2d")

    def forward(self, images, all_stages=False, want=DEFAULT_OUTPUTS, pipelined=False):
        """images [B,224,224,3] cuda float32 -> list (one dict per returned stage) of output tensors.
        pipelined=True: hpe_forward_pipelined (the outputs are complete after ``join()``); used by Predictor.predict for inputs
        larger than config.batch_size, whose chunks then overlap tail and encoder.  The ctx's tail stream writes the outputs and
        torch's caching allocator does not know that stream: keep the returned tensors alive until ``join()`` has been called
        (they are also marked with ``record_stream`` for the tail stream, so a tensor dropped early is not recycled under it)."""
        images = _require_cuda_tensor(images, "images", (224, 224, 3))
        B = images.shape[0]
        n_outs = self.num_stage if all_stages else 1
        outs = []
        arr = (_lib.HpeOutputs * n_outs)()
        for i in range(n_outs):
            t, o = self._alloc_outputs(B, want)
            outs.append(t)
            arr[i] = o
        fwd = self.lib.hpe_forward_pipelined if pipelined else self.lib.hpe_forward
        if pipelined:
            ts = self.tail_stream()
            for t in outs:
                for v in t.values():
                    v.record_stream(ts)
        _lib.check(fwd(self._h, images.data_ptr(), B, arr, n_outs, self._stream()))
        return outs

    def tail(self, features, all_stages=False, want=DEFAULT_OUTPUTS):
        """The regressor + SMPL half alone (hpe_tail): features [B,2048] from ``encoder()`` -> list of per-stage output dicts."""
        features = _require_cuda_tensor(features, "features", (2048,))
        B = features.shape[0]
        n_outs = self.num_stage if all_stages else 1
        outs = []
        arr = (_lib.HpeOutputs * n_outs)()
        for i in range(n_outs):
            t, o = self._alloc_outputs(B, want)
            outs.append(t)
            arr[i] = o
        _lib.check(self.lib.hpe_tail(self._h, features.data_ptr(), B, arr, n_outs, self._stream()))
        return outs

    def make_overlapped_plan(self, B, all_stages=False, want=DEFAULT_OUTPUTS, graph=False, tail_extra=None, n_sets=2):
        """Steady-state serving as ONE stream-ordered, hipGraph-capturable step per batch (hpe_encoder + hpe_tail):

            step(images_k)  =  fork;  side stream: tail(features of batch k-1) [+ tail_extra(outs)]  ||  encoder(images_k);  join

        i.e. the software pipeline of ``hpe_forward_pipelined`` with the overlap INSIDE the step instead of across calls, so that
        the whole step (every launch of both branches, the fork / join of the encoder's chunk streams and of the side stream) can
        be captured once and replayed with one host call.  ``step(images)`` returns the output set of the PREVIOUS batch (None
        on the first call); ``flush()`` runs the tail of the last batch alone and returns its outputs.  Features alternate
        between two buffers and outputs between ``n_sets`` sets; ``tail_extra(outs, set_index)`` (e.g. the loss call) is enqueued
        on the side stream after the tail, inside the capture.  graph=True: two graphs (one per feature buffer parity) per output
        set rotation are captured after an eager warm-up; images are copied into a static input buffer."""
        torch = _torch()
        n_outs = self.num_stage if all_stages else 1
        sets = []
        for _ in range(n_sets):
            outs, arr = [], (_lib.HpeOutputs * n_outs)()
            for i in range(n_outs):
                t, o = self._alloc_outputs(B, want)
                outs.append(t)
                arr[i] = o
            sets.append((outs, arr))
        feats = [self._new(B, 2048) for _ in range(2)]
        # the side branch runs on the ctx's own tail stream: a process should keep <= 4 busy HIP streams ("hardware queues")
        side = self.tail_stream()
        lib, h = self.lib, self._h
        state = {"k": 0}

        def enqueue(images, k, with_tail, with_enc):
            cur = torch.cuda.current_stream(self.tdev)
            if with_tail:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    outs, arr = sets[(k - 1) % n_sets]
                    _lib.check(lib.hpe_tail(h, feats[(k - 1) & 1].data_ptr(), B, arr, n_outs, C.c_void_p(side.cuda_stream)))
                    if tail_extra is not None:
                        tail_extra(outs, (k - 1) % n_sets)
            if with_enc:
                _lib.check(lib.hpe_encoder(h, images.data_ptr(), B, feats[k & 1].data_ptr(), self._stream()))
            if with_tail:
                cur.wait_stream(side)

        if not graph:

            def step(images):
                k = state["k"]
                enqueue(images, k, k > 0, True)
                state["k"] = k + 1
                if k > 0:
                    step.last = (k - 1) % n_sets
                return sets[(k - 1) % n_sets][0] if k > 0 else None

            def flush():
                k = state["k"]
                if k == 0:
                    return None
                enqueue(None, k, True, False)
                state["k"] = 0
                step.last = (k - 1) % n_sets
                return sets[(k - 1) % n_sets][0]

            step.flush = flush
            step.last = None  # index (into step.sets) of the output set of the most recently completed batch
            step.sets = [s_[0] for s_ in sets]
            return step

        static_in = torch.zeros((B, 224, 224, 3), dtype=torch.float32, device=self.tdev)
        self.enable_timing(0)  # event timing cannot be captured
        # eager warm-up of every launch shape (lazy module loading, workspace growth) before capture
        enqueue(static_in, 0, False, True)
        enqueue(static_in, 1, True, True)
        enqueue(None, 2, True, False)
        torch.cuda.synchronize(self.tdev)
        period = 2 * n_sets // (2 if n_sets % 2 == 0 else 1)  # lcm(2, n_sets): (feature parity, output set) repeats with this period

        def capture(k, with_tail, with_enc):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue(static_in, k, with_tail, with_enc)
            return g

        g_first = capture(0, False, True)
        g_steady = [capture(period + r, True, True) for r in range(period)]  # index r == k % period, k >= 1
        g_flush = [capture(period + r, True, False) for r in range(period)]

        def step(images):
            k = state["k"]
            static_in.copy_(images, non_blocking=True)
            (g_first if k == 0 else g_steady[k % period]).replay()
            state["k"] = k + 1
            if k > 0:
                step.last = (k - 1) % n_sets
            return sets[(k - 1) % n_sets][0] if k > 0 else None

        def flush():
            k = state["k"]
            if k == 0:
                return None
            g_flush[k % period].replay()
            state["k"] = 0
            step.last = (k - 1) % n_sets
            return sets[(k - 1) % n_sets][0]

        step.flush = flush
        step.last = None
        step.sets = [s_[0] for s_ in sets]
        step.graphs = (g_first, g_steady, g_flush)  # keep alive
        return step

    def tail_stream(self):
        """torch view of the ctx's tail stream (hpe_tail_stream): consumers of a pipelined plan's outputs enqueue there."""
        torch = _torch()
        ptr = self.lib.hpe_tail_stream(self._h)
        if not ptr:
            raise _lib.HpeError("no tail stream (ctx not finalized)")
        return torch.cuda.ExternalStream(ptr, device=self.tdev)

    def join(self):
        """Make the current stream wait for the tail of the last pipelined forward (hpe_join)."""
        _lib.check(self.lib.hpe_join(self._h, self._stream()))

    def make_forward_plan(self, B, all_stages=False, want=DEFAULT_OUTPUTS, graph=False, pipelined=False):
        """Pre-allocate outputs once; returns (callable(images), outputs) -- the steady-state serving path.
        graph=True captures the whole forward (all kernel launches, including the fork/join of the batch-chunk streams)
        into a hipGraph through torch.cuda.CUDAGraph: the callable then copies `images` into a static input buffer and
        replays the graph -- one host call per batch instead of ~75 launches (what matters for small batches).
        pipelined=True uses hpe_forward_pipelined (steady-state throughput: the tail of batch k overlaps the encoder of batch
        k+1); the caller reads the outputs after ``join()`` or from work enqueued on ``tail_stream()``."""
        torch = _torch()
        n_outs = self.num_stage if all_stages else 1
        outs = []
        arr = (_lib.HpeOutputs * n_outs)()
        for i in range(n_outs):
            t, o = self._alloc_outputs(B, want)
            outs.append(t)
            arr[i] = o
        lib, h = self.lib, self._h
        fwd = lib.hpe_forward_pipelined if pipelined else lib.hpe_forward
        if pipelined and graph:
            raise ValueError("a pipelined plan cannot be captured into a graph")

        def launch(images):
            # pipelined: encoder on the current stream, regressor + SMPL tail on the ctx's tail stream (outputs valid after
            # join(), or for work enqueued on tail_stream()); the next call's encoder overlaps this call's tail
            _lib.check(fwd(h, images.data_ptr(), B, arr, n_outs, self._stream()))

        if not graph:

            def run(images):
                launch(images)
                return outs

            return run, outs

        static_in = torch.zeros((B, 224, 224, 3), dtype=torch.float32, device=self.tdev)
        self.enable_timing(0)  # event timing cannot be captured
        side = torch.cuda.Stream(device=self.tdev)
        side.wait_stream(torch.cuda.current_stream(self.tdev))
        with torch.cuda.stream(side):  # warm-up outside capture (lazy allocations such as module loading)
            launch(static_in)
        torch.cuda.current_stream(self.tdev).wait_stream(side)
        torch.cuda.synchronize(self.tdev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launch(static_in)

        def run_graph(images):
            static_in.copy_(images, non_blocking=True)
            g.replay()
            return outs

        run_graph.graph = g  # keep alive
        return run_graph, outs

    def encoder(self, images):
        images = _require_cuda_tensor(images, "images", (224, 224, 3))
        feat = self._new(images.shape[0], 2048)
        _lib.check(self.lib.hpe_encoder(self._h, images.data_ptr(), images.shape[0], feat.data_ptr(), self._stream()))
        return feat

    def regress_stage(self, features, theta_prev=None):
        features = _require_cuda_tensor(features, "features", (2048,))
        B = features.shape[0]
        tp = None
        if theta_prev is not None:
            theta_prev = _require_cuda_tensor(theta_prev, "theta_prev", (85,))
            tp = theta_prev.data_ptr()
        out = self._new(B, 85)
        _lib.check(self.lib.hpe_regress_stage(self._h, features.data_ptr(), tp, B, out.data_ptr(), self._stream()))
        return out

    def smpl(self, theta, want=("verts", "joints", "J_transformed", "kp2d", "Rs")):
        theta = _require_cuda_tensor(theta, "theta", (85,))
        B = theta.shape[0]
        t, o = self._alloc_outputs(B, want)
        _lib.check(self.lib.hpe_smpl(self._h, theta.data_ptr(), B, C.byref(o), self._stream()))
        return t

    def mesh_loss(self, seg, verts2d):
        torch = _torch()
        seg = _require_cuda_tensor(seg, "seg")
        verts2d = _require_cuda_tensor(verts2d, "verts2d")
        B, H, W = seg.shape[0], seg.shape[1], seg.shape[2]
        P = verts2d.shape[1]
        out = torch.zeros(4, dtype=torch.float32, device=self.tdev)
        _lib.check(self.lib.hpe_mesh_loss(self._h, seg.data_ptr(), verts2d.data_ptr(), B, H, W, P, out.data_ptr(), self._stream()))
        return out[0]

    def val_losses(self, kp_gt, kp2d_stages, seg=None, verts2d_stages=None, out=None):
        """Both reprojection losses of every IEF stage in one call (hpe_val_losses): -> tensor [n_stage, 4] =
        (kp numerator, kp count, kp loss, mesh loss sum) per stage.  The silhouette-only work is done once per call."""
        torch = _torch()
        kp_gt = _require_cuda_tensor(kp_gt, "kp_gt")
        n = len(kp2d_stages)
        kp2d = [_require_cuda_tensor(t, "kp2d") for t in kp2d_stages]
        B, K = kp_gt.shape[0], kp_gt.shape[1]
        kp_ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in kp2d])
        H = W = P = 0
        seg_ptr, v_ptrs, keep = None, None, None
        if seg is not None and verts2d_stages is not None:
            seg = _require_cuda_tensor(seg, "seg")
            keep = [_require_cuda_tensor(t, "verts2d") for t in verts2d_stages]
            if len(keep) != n or seg.shape[0] != B:
                raise ValueError("one verts2d tensor per stage and one silhouette per image are needed")
            H, W, P = seg.shape[1], seg.shape[2], keep[0].shape[1]
            seg_ptr = seg.data_ptr()
            v_ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in keep])
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.tdev)
        _lib.check(self.lib.hpe_val_losses(self._h, seg_ptr, kp_gt.data_ptr(), kp_ptrs, v_ptrs, n, B, K, H, W, P, out.data_ptr(), self._stream()))
        return out

    def check_device(self):
        """Synchronise the current stream and raise if a kernel flagged an invalid result (hpe_device_status)."""
        _lib.check(self.lib.hpe_device_status(self._h, self._stream()))

    def debug_conv(self, idx, x, residual=None, relu=True):
        s = CONV_SPECS[idx]
        x = _require_cuda_tensor(x, "x")
        B = x.shape[0]
        y = self._new(B, s.hout, s.hout, s.cout)
        r = None
        if residual is not None:
            r = _require_cuda_tensor(residual, "residual").data_ptr()
        _lib.check(self.lib.hpe_debug_conv(self._h, idx, x.data_ptr(), B, r, int(relu), y.data_ptr(), self._stream()))
        return y

    def debug_chain(self, idx2c, t2, residual):
        """bf16 contexts: res*_branch2c (+ residual + ReLU) and the next block's res*_branch2a (+ ReLU) as the one launch of
        conv_chain_bf16.hip.  Returns (t3 [B,H,H,4C], u1 [B,H,H,C], resident workgroups per CU of the two instantiations)."""
        s = CONV_SPECS[idx2c]
        first = CONV_SPECS[idx2c + 1].name.endswith("branch1")  # conv_block form: `residual` is the block input
        sn = CONV_SPECS[idx2c + (2 if first else 1)]
        t2 = _require_cuda_tensor(t2, "t2", (s.hin, s.hin, s.cin))
        residual = _require_cuda_tensor(residual, "residual", (s.hout, s.hout, CONV_SPECS[idx2c + 1].cin if first else s.cout))
        B = t2.shape[0]
        t3 = self._new(B, s.hout, s.hout, s.cout)
        u1 = self._new(B, sn.hout, sn.hout, sn.cout)
        occ = (C.c_int * 3)()
        _lib.check(self.lib.hpe_debug_chain(self._h, idx2c, t2.data_ptr(), residual.data_ptr(), B, t3.data_ptr(), u1.data_ptr(), occ, self._stream()))
        return t3, u1, (occ[0], occ[1], occ[2])

    def debug_stem(self, images, rows_per_strip=0):
        images = _require_cuda_tensor(images, "images", (224, 224, 3))
        y = self._new(images.shape[0], 56, 56, 64)
        _lib.check(self.lib.hpe_debug_stem(self._h, images.data_ptr(), images.shape[0], int(rows_per_strip), y.data_ptr(), self._stream()))
        return y

    def joint_regress(self, X, use_kp_regressor=True):
        X = _require_cuda_tensor(X, "X", (NUM_VERTS, 3))
        K = self.num_kp if use_kp_regressor else 24
        out = self._new(X.shape[0], K, 3)
        _lib.check(self.lib.hpe_debug_joint_regress(self._h, X.data_ptr(), X.shape[0], int(use_kp_regressor), out.data_ptr(), self._stream()))
        return out

    def encoder_kernel_description(self):
        """The kernel family bench.py's `roofline` block prices (one string per encoder dtype, kept next to the dispatch)."""
        if self.encoder_dtype == "fp32":
            return ("conv_gemm_f32_dma_kernel (the 1x1 / strided / dual-source layers) + w4_input_kernel + w4_gemm_kernel / w4_gemm32_kernel "
                    "(the 13 3x3 layers on the 28x28 / 14x14 / 7x7 maps as fp32 Winograd F(4x4,3x3); F(2x2,3x3) / direct below 64 work "
                    "items) + wino_fused_kernel (the three 56x56 3x3 layers, F(2x2,3x3)) + stem_fused_f32_kernel -- the 53 conv layers of one "
                    "step, priced at their direct-convolution FLOPs")
        return ("conv_gemm_bf16_dma_kernel (1x1 / strided / dual-source layers) + chain_expand_reduce_bf16_kernel (branch2c + next branch2a of "
                "stages 2-3 as one launch) + conv3_halo_bf16_kernel (the sixteen 3x3 layers, tile + halo resident in LDS) + stem_fused_bf16_kernel "
                "-- the 53 conv layers of one step, priced at their algorithmic HBM bytes")

    def enable_timing(self, level=1):
        _lib.check(self.lib.hpe_enable_timing(self._h, int(level)))

    def timings(self):
        ms = (C.c_float * 5)()
        _lib.check(self.lib.hpe_get_timings(self._h, ms))
        return dict(encoder_ms=ms[0], conv_ms=ms[1], regress_smpl_ms=ms[2], total_ms=ms[4])

    def span_stats(self):
        """Encoder span over all timed calls since enable_timing (hpe_get_span_stats): dict(mean_ms, min_ms, max_ms, calls)."""
        ms = (C.c_float * 3)()
        n = C.c_int()
        _lib.check(self.lib.hpe_get_span_stats(self._h, ms, C.byref(n)))
        return dict(mean_ms=ms[0], min_ms=ms[1], max_ms=ms[2], calls=n.value)

    def set_loss_counter(self, counter=None):
        """counter: int64 CUDA tensor of 2 elements (zeroed by the caller) that the pixel -> vertex searches add their MFMA counts
        to ([0] cell-grid search, [1] full search; 1024 (pixel, vertex) pairs per MFMA); None disables."""
        _lib.check(self.lib.hpe_debug_set_loss_counter(self._h, None if counter is None else counter.data_ptr()))
        self._loss_counter = counter  # keep alive

    def loss_timings(self):
        ms = (C.c_float * 2)()
        _lib.check(self.lib.hpe_get_loss_timings(self._h, ms))
        return dict(val_losses_ms=ms[0], a2b_search_ms=ms[1])

    def conv_timings(self):
        ms = (C.c_float * _lib.NUM_CONV)()
        _lib.check(self.lib.hpe_get_conv_timings(self._h, ms))
        return list(ms)


# ---------------------------------------------------------------------- context-free operators
def _cur_stream(t):
    return C.c_void_p(_torch().cuda.current_stream(t.device).cuda_stream)


def orth_proj(X, camera):
    X = _require_cuda_tensor(X, "X")
    camera = _require_cuda_tensor(camera, "camera").reshape(-1, 3)
    B, P = X.shape[0], X.shape[1]
    out = _torch().empty((B, P, 2), dtype=_torch().float32, device=X.device)
    with _torch().cuda.device(X.device):
        _lib.check(_lib.load().hpe_orth_proj(X.data_ptr(), camera.data_ptr(), B, P, out.data_ptr(), _cur_stream(X)))
    return out


def reproject(verts, cam, im_w, im_h):
    verts = _require_cuda_tensor(verts, "verts")
    cam = _require_cuda_tensor(cam, "cam").reshape(-1, 3)
    B, P = verts.shape[0], verts.shape[1]
    out = _torch().empty((B, P, 2), dtype=_torch().float32, device=verts.device)
    with _torch().cuda.device(verts.device):
        _lib.check(
            _lib.load().hpe_reproject_vertices(verts.data_ptr(), cam.data_ptr(), B, P, float(im_w), float(im_h), out.data_ptr(), _cur_stream(verts))
        )
    return out


def kp_loss_parts(kp_gt, kp_pred):
    """-> tensor [3]: (sum vis*|d|, 2*#visible, loss)"""
    kp_gt = _require_cuda_tensor(kp_gt, "kp_gt")
    kp_pred = _require_cuda_tensor(kp_pred, "kp_pred")
    B, K = kp_gt.shape[0], kp_gt.shape[1]
    out = _torch().zeros(4, dtype=_torch().float32, device=kp_gt.device)
    with _torch().cuda.device(kp_gt.device):
        _lib.check(_lib.load().hpe_kp_loss(kp_gt.data_ptr(), kp_pred.data_ptr(), B, K, out.data_ptr(), _cur_stream(kp_gt)))
    return out[:3]
