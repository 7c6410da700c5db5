"""MI355X execution engine of the NOVA generation hot path.

Drives libnova_hip.so (C ABI in include/nova_hip.h) for everything below
`Transformer3DModel.forward` in eval mode (reference diffnext/models/transformers/
transformer_3d.py:102-164,192-200): text/condition prefix, the 16-block conditioning ViT, the
masked-autoregressive loop over the 32-block ViT (known-token gather -> first half -> scatter ->
second half -> final LN on the rows being predicted) and the flow-matching denoise loop of the
diffusion MLP. PyTorch supplies device memory, the current stream and a few index tensors; all
arithmetic on activations runs in the hand-written gfx950 kernels. There is no fallback: a
missing library or a non-gfx950 device raises `NovaHipError`.

Data layout in HBM (all row-major, one token per row):
  weights      GEMM weights in the activation dtype exactly as nn.Linear stores them ([N][K]);
               biases / LayerNorm affines as float32; AdaLN projections of all diffusion blocks
               concatenated to one [(3*depth+2)D, D] matrix
  activations  x [S*L, D] residual stream (S = 2B guidance rows: cond then uncond), fused
               qkv [S*L, 3D] read in place by attention, hidden [S*L, 4D]
  point state  canvas [B, N, P] float32 patch vectors (P = p*p*C, = xyz for point sets),
               mask [B, N] float32, generation order [B, N] int64
"""
import ctypes
import os

import numpy as np
import torch

from . import hip

_F32 = torch.float32
_POISON_WS = os.environ.get("NOVA_POISON_WS", "0") == "1"  # fill freshly allocated lane workspaces with NaN / -1 (debugging aid)
# default number of half-batch lanes; read ONCE at import (an experiment knob of tools/lanes_ab.py, 0 = the rule below)
_ENV_LANES = int(os.environ.get("NOVA_LANES", "0") or 0)


def _f32(t):
    return t.detach().to(_F32).contiguous()


class _Pack(object):
    """Device-resident parameter views/copies + the ctypes structs pointing at them."""

    def __init__(self):
        self.keep = []

    def w(self, t, dtype):
        t = t.detach()
        t = t if (t.dtype == dtype and t.is_contiguous()) else t.to(dtype).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def f(self, t):
        t = _f32(t)
        self.keep.append(t)
        return t.data_ptr()


def pack_vit_blocks(blocks, dtype):
    """nova_vit_block[] for a list of `Block` modules (state_dict names in include/nova_hip.h)."""
    pk = _Pack()
    arr = (hip.VitBlock * len(blocks))()
    for i, b in enumerate(blocks):
        arr[i] = hip.VitBlock(
            pk.w(b.attn.qkv.weight, dtype), pk.f(b.attn.qkv.bias), pk.w(b.attn.proj.weight, dtype), pk.f(b.attn.proj.bias),
            pk.f(b.norm1.weight), pk.f(b.norm1.bias), pk.w(b.mlp.fc1.weight, dtype), pk.f(b.mlp.fc1.bias),
            pk.w(b.mlp.fc2.weight, dtype), pk.f(b.mlp.fc2.bias), pk.f(b.norm2.weight), pk.f(b.norm2.bias))
    pk.arr = arr
    pk.modules = list(blocks)
    return pk


def pack_vit_blocks_fp8(blocks):
    """nova_vit_block_fp8[]: the fused-QKV, fc1 and fc2 weights of every block as OCP e4m3 bytes + one scale per output row
    (amax / 448), quantised once here from the bf16 parameters (BASELINE configs[4]; the reference has no fp8 path)."""
    pk = _Pack()
    arr = (hip.VitBlockFp8 * len(blocks))()
    for i, b in enumerate(blocks):
        fields = []
        for lin in (b.attn.qkv, b.mlp.fc1, b.mlp.fc2):
            w8, ws = hip.quantize_rows_fp8(lin.weight.detach().to(torch.bfloat16).contiguous())
            pk.keep += [w8, ws]
            fields += [w8.data_ptr(), ws.data_ptr()]
        arr[i] = hip.VitBlockFp8(*fields)
    pk.arr = arr
    return pk


def patch_weight(conv):
    """Conv2d [D, C, p, p] -> [D, p*p*C] in the patchified (row, col, channel) order."""
    w = conv.weight.detach()
    return w.permute(0, 2, 3, 1).reshape(w.size(0), -1).contiguous()


def pack_decoder(dec, dtype):
    """nova_decoder for a `DiffusionMLP` module."""
    pk = _Pack()
    depth = len(dec.blocks)
    blocks = (hip.MlpBlock * depth)()
    for i, b in enumerate(dec.blocks):
        blocks[i] = hip.MlpBlock(pk.w(b.proj.fc1.weight, dtype), pk.f(b.proj.fc1.bias), pk.w(b.proj.fc2.weight, dtype),
                                 pk.f(b.proj.fc2.bias), pk.f(b.norm2.weight), pk.f(b.norm2.bias))
    adaln_w = torch.cat([b.norm1.proj.weight.detach() for b in dec.blocks] + [dec.norm.proj.weight.detach()])
    adaln_b = torch.cat([b.norm1.proj.bias.detach() for b in dec.blocks] + [dec.norm.proj.bias.detach()])
    pk.blocks = blocks
    pk.struct = hip.Decoder(depth, ctypes.cast(blocks, ctypes.POINTER(hip.MlpBlock)), pk.w(adaln_w, dtype), pk.f(adaln_b),
                            pk.w(patch_weight(dec.patch_embed.proj), dtype), pk.f(dec.patch_embed.proj.bias),
                            pk.w(dec.head.weight, dtype), pk.f(dec.head.bias))
    tc = dec.time_cond_embed
    pk.time = [(pk.w(p.fc1.weight, dtype), pk.f(p.fc1.bias), pk.w(p.fc2.weight, dtype), pk.f(p.fc2.bias))
               for p in (tc.timestep_proj, tc.condition_proj)]
    pk.depth = depth
    return pk


class _Guidance(object):
    """The guidance knobs of guidance_scaler.py:24-44 the loop needs (scale decay, truncation, renorm)."""

    def __init__(self, inputs):
        self.guidance_scale = inputs.get("guidance_scale", 1)
        self.guidance_trunc = inputs.get("guidance_trunc", 0)
        self.guidance_renorm = inputs.get("guidance_renorm", 1)
        self.image_guidance_scale = inputs.get("image_guidance_scale", 0) or 0
        self.spatiotemporal_guidance_scale = inputs.get("spatiotemporal_guidance_scale", 0) or 0
        self.extra_pass = self.image_guidance_scale + self.spatiotemporal_guidance_scale > 0
        self.min_guidance_scale = inputs.get("min_guidance_scale", None) or self.guidance_scale
        self.inc_guidance_scale = self.guidance_scale - self.min_guidance_scale

    def decay_guidance_scale(self, decay=0):
        self.guidance_scale = self.inc_guidance_scale * decay + self.min_guidance_scale


def sampler_plan(sched, steps):
    """(timesteps f32[steps], per-step (kx, kv, clip, c0, cx, sigma), ancestral?) for `nova_sampler_step`:
    x0 = clamp(kx x + kv v); x <- c0 x0 + cx x + sigma noise."""
    kind = type(sched).__name__
    sched.set_timesteps(steps)
    if kind == "FlowMatchEulerDiscreteScheduler":  # scheduling_cfm.py:92-104,134-136
        sig = sched.sigmas
        return np.asarray(sched.timesteps, dtype="float32"), [(0.0, 1.0, 0.0, sig[j + 1] - sig[j], 1.0, 0.0) for j in range(steps)], False
    if kind == "DDPMScheduler":  # scheduling_ddpm.py:236-316 (no clipping / thresholding stage in the reference's step)
        if sched.variance_type not in ("fixed_small", "fixed_small_log", "fixed_large"):
            raise NotImplementedError(f"DDPM variance_type {sched.variance_type} is not built on the HIP path")
        ts = [int(t) for t in sched.timesteps]
        coefs = []
        for t in ts:
            prev_t = int(sched.previous_timestep(t))
            a_t = float(sched.alphas_cumprod[t])
            a_prev = float(sched.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
            b_t, b_prev = 1 - a_t, 1 - a_prev
            cur_alpha = a_t / a_prev
            cur_beta = 1 - cur_alpha
            kind_p = sched.config.prediction_type
            if kind_p == "epsilon":
                kx, kv = a_t ** -0.5, -(b_t ** 0.5) / a_t ** 0.5
            elif kind_p == "sample":
                kx, kv = 0.0, 1.0
            elif kind_p == "v_prediction":
                kx, kv = a_t ** 0.5, -(b_t ** 0.5)
            else:
                raise ValueError(f"Unsupported prediction type given as {kind_p}.")
            var = max(b_prev / b_t * cur_beta, 1e-20)
            sigma = 0.0 if t <= 0 else (cur_beta ** 0.5 if sched.variance_type == "fixed_large" else var ** 0.5)
            coefs.append((kx, kv, 0.0, a_prev ** 0.5 * cur_beta / b_t, cur_alpha ** 0.5 * b_prev / b_t, sigma))
        return np.asarray(ts, dtype="float32"), coefs, True
    raise NotImplementedError(f"sampler {kind} is not built on the HIP path (flow-matching Euler and DDPM are)")


def _params_signature(module):
    return tuple((p.data_ptr(), p._version, p.dtype) for p in module.parameters())


class NovaEngine(object):
    """Runs `Transformer3DModel` generation on one MI355X. Built lazily per model, re-packed when
    parameters move, change dtype or are updated in place."""

    def __init__(self, model):
        hip.load()
        self.model = model
        self.sig = None
        self.fp8 = False
        self.fp8_delayed = True
        self.ws = {}  # lane -> {shape key: buffers} in order of last use (see _workspace)

    # ------------------------------------------------------------------ packing / workspaces
    @classmethod
    def for_model(cls, model):
        eng = model.__dict__.get("_nova_engine", None)
        if eng is None:
            eng = model.__dict__["_nova_engine"] = cls(model)
        return eng

    def _refresh(self):
        m = self.model
        sig = _params_signature(m)
        if sig == self.sig:
            return
        dev = next(m.parameters()).device
        dtype = next(p for p in m.parameters() if p.is_floating_point()).dtype
        hip.dtype_code(dtype)  # raises for anything but f32 / bf16
        self.dev, self.dtype, self.code = dev, dtype, hip.dtype_code(dtype)
        ve, ie, de = m.video_encoder, m.image_encoder, m.image_decoder
        self.D = ie.embed_dim
        self.heads = ie.blocks[0].attn.num_heads
        self.hidden = ie.blocks[0].mlp.fc1.out_features
        self.P = ie.patch_embed.patch_size ** 2 * ie.image_dim  # values per point token (xyz for point sets)
        self.video = pack_vit_blocks(list(ve.blocks), dtype)
        half = ie.encoder_depth
        self.enc1 = pack_vit_blocks(list(ie.blocks[:half]), dtype)
        self.enc2 = pack_vit_blocks(list(ie.blocks[half:]), dtype)
        # The LAST encoder block is only consumed at the rows predicted in this AR step (final LN + decoder read
        # nothing else), so everything but its K/V projection runs on those n rows only (same math per row).
        self.enc2_head = pack_vit_blocks(list(ie.blocks[half:-1]), dtype) if len(ie.blocks) - half > 1 else None
        lb = ie.blocks[-1]
        lpk = self.last = _Pack()
        W, Bq = lb.attn.qkv.weight.detach(), lb.attn.qkv.bias.detach()
        Dm = W.shape[1]
        lpk.q = (lpk.w(W[:Dm], dtype), lpk.f(Bq[:Dm]))
        lpk.kv = (lpk.w(W[Dm:], dtype), lpk.f(Bq[Dm:]))
        lpk.proj = (lpk.w(lb.attn.proj.weight, dtype), lpk.f(lb.attn.proj.bias))
        lpk.fc1 = (lpk.w(lb.mlp.fc1.weight, dtype), lpk.f(lb.mlp.fc1.bias))
        lpk.fc2 = (lpk.w(lb.mlp.fc2.weight, dtype), lpk.f(lb.mlp.fc2.bias))
        lpk.n1 = (lpk.f(lb.norm1.weight), lpk.f(lb.norm1.bias))
        lpk.n2 = (lpk.f(lb.norm2.weight), lpk.f(lb.norm2.bias))
        self.dec = pack_decoder(de, dtype)
        pk = self.misc = _Pack()
        te = m.text_embed  # None: a model conditioned through pre-supplied rows only (inputs["c"], transformer_3d.py:66)
        self.text = None if te is None else (pk.w(te.proj.weight, dtype), pk.f(te.proj.bias), pk.f(te.norm.weight), pk.f(te.norm.bias))
        self.vnorm = (pk.f(ve.norm.weight), pk.f(ve.norm.bias))
        self.inorm = (pk.f(ie.norm.weight), pk.f(ie.norm.bias))
        self.patch = (pk.w(patch_weight(ie.patch_embed.proj), dtype), pk.f(ie.patch_embed.proj.bias))
        self.vpatch = (pk.w(patch_weight(ve.patch_embed.proj), dtype), pk.f(ve.patch_embed.proj.bias))  # frames t > 0
        self.mask_token = pk.w(m.mask_embed.mask_token, dtype)
        self.time_freq = torch.arange(128, dtype=_F32, device=dev).mul(-9.210340371976184 / 128).exp()
        self.vtime = None
        vpe = m.video_pos_embed
        if hasattr(vpe, "time_proj"):  # abs-PE checkpoints: VideoPosEmbed's frame-index MLP (embeddings.py:94-111)
            self.vtime = (pk.w(vpe.time_proj[0].weight, dtype), pk.f(vpe.time_proj[0].bias), pk.w(vpe.time_proj[2].weight, dtype),
                          pk.f(vpe.time_proj[2].bias), pk.f(vpe.norm.weight), pk.f(vpe.norm.bias))
        self.motion = None
        if m.motion_embed is not None:  # MotionEmbed's (flow, fps) MLPs (embeddings.py:119-137)
            self.motion = [(pk.w(pr[0].weight, dtype), pk.f(pr[0].bias), pk.w(pr[2].weight, dtype), pk.f(pr[2].bias))
                           for pr in (m.motion_embed.flow_proj, m.motion_embed.fps_proj)]
        self.sig = sig

    # AdaLN projections of all diffusion steps in one GEMM (nova_decoder_denoise mod_steps = steps) when the steps-times
    # larger modulation buffer stays under this many bytes per lane (config C: 3.3 GB; 288 GB of HBM per GPU)
    MOD_HOIST_BYTES = 12 << 30

    # A lane keeps the scratch buffers of its last WS_KEEP call shapes (most recent last), as long as all kept buffers stay under
    # WS_KEEP_FRACTION of the device memory: a serving loop that alternates between a few batch sizes then neither re-allocates nor
    # re-captures its denoising-loop graphs (their keys hold these buffers' addresses) on every change. Evicting a set drops the graph
    # cache (nova_debug_drop_graphs waits for the device first: a launch of one of those graphs may still be queued).
    WS_KEEP = 3
    WS_KEEP_FRACTION = 0.25

    def _workspace(self, S, B, N, L, nmax, lane=0, steps=1):
        """Scratch buffers of one lane for this call shape (dtype / fp8 mode included): allocated on first use, kept as described above."""
        key = (S, B, N, L, nmax, self.dtype, self.dev, self.fp8, steps)
        kept = self.ws.setdefault(lane, {})  # shape key -> buffers, in order of last use
        if key in kept:
            kept[key] = kept.pop(key)  # most recently used last
        else:
            D, dt, dev = self.D, self.dtype, self.dev
            e = lambda *shape: torch.empty(*shape, dtype=dt, device=dev)
            rows = S * L
            ws = dict(x8=torch.empty(rows, D, dtype=torch.uint8, device=dev), xs=torch.empty(rows, dtype=_F32, device=dev),
                      h8=torch.empty(rows, self.hidden, dtype=torch.uint8, device=dev), hs=torch.empty(rows, dtype=_F32, device=dev)) if self.fp8 else {}
            ws.update(x1=e(rows, D), x2=e(rows, D), qkv=e(rows, 3 * D), a=e(rows, D), b=e(rows, D),
                      h=e(rows, self.hidden), z0=e(B * N, D), da=e(steps * S * nmax, D), du=e(S * nmax, D), dh=e(S * nmax, D),
                      df=e(S * nmax, D), dg=e(S * nmax, D), dmod=e(steps * S * nmax, (3 * self.dec.depth + 2) * D),
                      # operands of nova_decoder_denoise at addresses that do not change from call to call (its launch
                      # sequence is replayed as a hipGraph keyed by its arguments): condition rows and the rows being denoised
                      dz=e(S * nmax, D), dx=torch.empty(B * nmax * self.P, dtype=_F32, device=dev), temb={},
                      # per-AR-step temporaries of the hot loop (flat, viewed at the step's sizes): no allocator call per step
                      ids_prev=torch.empty(B * N, dtype=torch.int64, device=dev), ids_pred=torch.empty(B * nmax, dtype=torch.int64, device=dev),
                      ids_cat=torch.empty(S * nmax, dtype=torch.int64, device=dev), rope1=torch.empty(B * L * (D // self.heads), dtype=_F32, device=dev),
                      rope_q=torch.empty(B * nmax * (D // self.heads), dtype=_F32, device=dev), lq=e(S * nmax, D), lx=e(S * nmax, D), lo=e(S * nmax, D),
                      lh=e(S * nmax, self.hidden), lz=e(S * nmax, D), lz2=e(S * nmax, D))
            if _POISON_WS:  # debugging aid: a read of a never-written scratch element shows as NaN instead of depending on what the
                for v in ws.values():  # allocator handed out (NOVA_POISON_WS=1, read once at import)
                    if torch.is_tensor(v) and v.is_floating_point():
                        v.fill_(float("nan"))
                    elif torch.is_tensor(v):
                        v.fill_(-1 if v.dtype == torch.int64 else 0xFF)
            kept[key] = ws
            self._evict_workspaces(lane, key)
        return kept[key]

    def _evict_workspaces(self, lane, keep_key):
        nbytes = lambda ws: sum(v.numel() * v.element_size() for v in ws.values() if torch.is_tensor(v))
        budget = self.WS_KEEP_FRACTION * torch.cuda.get_device_properties(self.dev).total_memory
        dropped = False
        kept = self.ws[lane]
        while len(kept) > 1 and (len(kept) > self.WS_KEEP or sum(nbytes(w) for lanes in self.ws.values() for w in lanes.values()) > budget):
            oldest = next(k for k in kept if k != keep_key)
            del kept[oldest]
            dropped = True
        if dropped:  # graphs captured for the evicted buffers' addresses can never be replayed again
            hip.call("nova_debug_drop_graphs")

    # ------------------------------------------------------------------ building blocks
    def _blocks(self, pack, x, S, L, rope, rope_batch, ws):
        if self.fp8 and L >= 16:  # QKV / fc1 / fc2 on the block-scaled fp8 MFMA (inputs["gemm_dtype"] == "fp8")
            q = pack.__dict__.get("fp8")
            if q is None:
                q = pack.fp8 = pack_vit_blocks_fp8(pack.modules)
            # delayed scaling of the MLP hidden rows (per lane and stack): the scale of this call comes from the largest |GELU|
            # the previous call of the same stack saw (x 2 headroom; e4m3 saturates beyond); the first call guesses 64 / 448
            state = ws.setdefault("q8", {}).get(pack)  # keyed by the pack object (kept alive by the key), not by a reusable id()
            if state is None and self.fp8_delayed:
                n = len(pack.arr)
                state = ws["q8"][pack] = (torch.full((n,), 64.0 / 448.0, dtype=_F32, device=self.dev),
                                              torch.zeros(n, dtype=torch.int32, device=self.dev))
            sc, am = state if self.fp8_delayed else (None, None)
            hip.call("nova_vit_blocks_forward_fp8", pack.arr, q.arr, len(pack.arr), x.data_ptr(), S, L, self.D, self.heads, self.hidden,
                     hip.ptr(rope), rope_batch, ws["qkv"].data_ptr(), ws["a"].data_ptr(), ws["b"].data_ptr(), ws["h"].data_ptr(),
                     ws["x8"].data_ptr(), ws["xs"].data_ptr(), ws["h8"].data_ptr(), ws["hs"].data_ptr(), hip.ptr(sc), hip.ptr(am),
                     hip.stream_ptr())
            if sc is not None:
                seen = am.view(_F32)
                torch.where(seen > 0, seen * (2.0 / 448.0), sc, out=sc)
                am.zero_()
            return
        hip.call("nova_vit_blocks_forward", pack.arr, len(pack.arr), x.data_ptr(), S, L, self.D, self.heads, self.hidden,
                 hip.ptr(rope), rope_batch, ws["qkv"].data_ptr(), ws["a"].data_ptr(), ws["b"].data_ptr(),
                 ws["h"].data_ptr(), self.code, hip.stream_ptr())

    def _gemm(self, a, w_ptr, b_ptr, N, act=hip.ACT_NONE, out=None):
        M, K = a.shape
        out = torch.empty(M, N, dtype=a.dtype, device=a.device) if out is None else out
        hip.call("nova_gemm_bias_act", a.data_ptr(), w_ptr, b_ptr, out.data_ptr(), M, N, K, act, self.code, hip.stream_ptr())
        return out

    def _norm_rows(self, x, gb, gather=None, rows=None, eps=1e-5, out=None):
        rows = (gather.numel() if gather is not None else x.shape[0]) if rows is None else rows
        out = torch.empty(rows, self.D, dtype=x.dtype, device=x.device) if out is None else out
        hip.call("nova_row_norm", x.data_ptr(), out.data_ptr(), gb[0], gb[1], None, 0, -1, -1, -1, None, hip.ptr(gather),
                 rows, self.D, eps, self.code, hip.stream_ptr())
        return out

    def _sequence(self, out, prefix, prefix_rows, tokens, tok_rows, ids, S, B, Lp, n_sel):
        hip.call("nova_build_sequence", hip.ptr(prefix) if Lp else None, prefix_rows, hip.ptr(tokens) if n_sel else None,
                 tok_rows, hip.ptr(ids), out.data_ptr(), S, B, Lp, n_sel, self.D, self.code, hip.stream_ptr())

    def _last_block_rows(self, x2, S, B, L, Nv, n, pred_ids, pos_img, inv_freq, rope_full, hd, ws):
        """Block.forward (vision_transformer.py:89-92) of the last encoder block for the n predicted rows only:
        K/V projected for all L rows, Q / attention / proj / LN / MLP for [S*n] gathered rows. Returns y [S*n, D]."""
        D, code, st, lp = self.D, self.code, hip.stream_ptr, self.last
        es = x2.element_size()
        # K,V for every token: x2 [S*L, D] x Wkv^T -> kv [S*L, 2D], RoPE on the K half
        kv = ws["qkv"].view(-1)[: S * L * 2 * D].view(S * L, 2 * D)
        hip.call("nova_qkv_rope_cols", x2.data_ptr(), lp.kv[0], lp.kv[1], hip.ptr(rope_full), kv.data_ptr(), S * L, 2 * D, D,
                 L, 1, hd, D, code, st())
        # the predicted rows of every sequence (cond and uncond share pred_ids); every temporary is a workspace slot
        rows = S * n
        ids = ws["ids_cat"][:rows].view(S, n)  # [S, n]: every guidance pass predicts the same tokens
        ids.view(S // B, B, n).copy_(pred_ids)
        xq, q, o, a = (ws[k][:rows] for k in ("lx", "lq", "lo", "lz"))
        hip.call("nova_build_sequence", None, 0, x2.data_ptr() + Nv * D * es, L, ids.data_ptr(), xq.data_ptr(), S, S, 0, n, D,
                 code, st())
        rope_q = hip.rope_table(pos_img, pred_ids, 0, inv_freq, B, hd, out=ws["rope_q"]) if pos_img is not None else None
        hip.call("nova_qkv_rope_cols", xq.data_ptr(), lp.q[0], lp.q[1], hip.ptr(rope_q), q.data_ptr(), S * n, D, D, n,
                 B if rope_q is not None else 1, hd, D, code, st())
        hip.call("nova_attn_fwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + D * es, o.data_ptr(), S, self.heads, n, L, hd,
                 D, 2 * D, D, float(hd) ** -0.5, code, st())
        self._gemm(o, lp.proj[0], lp.proj[1], D, out=a)
        hip.call("nova_row_norm", a.data_ptr(), xq.data_ptr(), lp.n1[0], lp.n1[1], None, 0, -1, -1, -1, xq.data_ptr(), None, S * n,
                 D, 1e-5, code, st())
        h = self._gemm(self._gemm(xq, lp.fc1[0], lp.fc1[1], self.hidden, hip.ACT_GELU_ERF, out=ws["lh"][:rows]), lp.fc2[0], lp.fc2[1], D, out=a)
        hip.call("nova_row_norm", h.data_ptr(), xq.data_ptr(), lp.n2[0], lp.n2[1], None, 0, -1, -1, -1, xq.data_ptr(), None, S * n,
                 D, 1e-5, code, st())
        return xq

    def timestep_table(self, timesteps):
        """temb[i] = timestep_proj(freq_embed(t_i)) for every diffusion step (diffusion_mlp.py:65-73)."""
        t = torch.as_tensor(np.asarray(timesteps, dtype="float32"), device=self.dev)
        feats = torch.empty(t.numel(), 256, dtype=self.dtype, device=self.dev)
        hip.call("nova_timestep_freq", t.data_ptr(), self.time_freq.data_ptr(), feats.data_ptr(), t.numel(), 256, self.code,
                 hip.stream_ptr())
        w1, b1, w2, b2 = self.dec.time[0]
        return self._gemm(self._gemm(feats, w1, b1, self.D, hip.ACT_SILU), w2, b2, self.D)

    # ------------------------------------------------------------------ the generation loop
    @torch.no_grad()
    def generate(self, inputs):
        """Eval-mode `Transformer3DModel.forward` body (transformer_3d.py:63-77,102-164,192-200). Returns x [B,C,T,H,W]
        (T = max_latent_length; a point set is the T = 1 case).

        The batch may be run as two half-batch LANES on two HIP streams (`inputs["lanes"]`, default 2 for B >= 4):
        samples are independent, so while one lane is in its latency-bound denoise loop (hundreds of small launches)
        the other lane's encoder GEMMs fill the idle CUs. All random draws stay here, for the whole batch and in the
        reference's order, and the lanes receive row slices - results do not depend on the number of lanes.
        """
        self._refresh()
        gemm_dtype = inputs.get("gemm_dtype", None)
        if gemm_dtype not in (None, "bf16", "fp8"):
            raise ValueError(f"gemm_dtype {gemm_dtype!r}: the encoder GEMMs run in the model dtype or in 'fp8'")
        self.fp8 = gemm_dtype == "fp8"
        self.fp8_delayed = not inputs.get("fp8_row_scaled_hidden", False)  # False: per-row quantisation pass between fc1 and fc2
        if self.fp8 and (self.dtype != torch.bfloat16 or self.D % 256 or self.hidden % 256):
            raise NotImplementedError("gemm_dtype='fp8' needs a bfloat16 model whose width and MLP width are multiples of 256")
        with torch.cuda.device(self.dev):  # launches go to the model's device whatever the caller's current device is
            return self._generate(inputs)

    def _generate(self, inputs):
        m = self.model
        dev, dtype = self.dev, self.dtype
        scaler = _Guidance(inputs)
        if scaler.image_guidance_scale and scaler.spatiotemporal_guidance_scale:
            raise ValueError("image_guidance_scale and spatiotemporal_guidance_scale are exclusive (the reference's expand_text "
                             "builds four text blocks for three guidance passes when both are set, guidance_scaler.py:46-57)")
        # Condition rows (transformer_3d.py:63-77): the caller's own list inputs["c"] (model-width rows [S0, Lc_i, D], e.g. label
        # embeddings), then TextEmbed(prompt) when the model has a text embedding and a prompt is given, then the motion tokens -
        # concatenated along the token axis in that order.
        pre = [t for t in (inputs.get("c", None) or [])]
        prompt = inputs.get("prompt", None)
        use_text = prompt is not None and m.text_embed is not None
        if not use_text and not pre:
            raise ValueError("no condition rows: give a prompt (model with a text embedding) or a list of rows inputs['c']")
        T = int(inputs.get("max_latent_length", 1))
        ie, ve = m.image_encoder, m.video_encoder
        C, (H, W), p = ie.image_dim, ie.image_size, ie.patch_embed.patch_size
        h, w = H // p, W // p
        N, P = h * w, p * p * C
        if not use_text:
            prompt = None
        elif isinstance(prompt, (tuple, list)):  # strings or per-prompt embeddings: host-side padding (embeddings.py:179-201)
            prompt = m.text_embed.encode_prompts(prompt)
        pre_c = None
        if pre:
            pre_c = (torch.cat(pre, dim=1) if len(pre) > 1 else pre[0]).to(device=dev, dtype=dtype)
            if pre_c.dim() != 3 or pre_c.shape[-1] != self.D or (prompt is not None and pre_c.shape[0] != prompt.shape[0]):
                raise ValueError(f"inputs['c'] rows {tuple(pre_c.shape)} do not fit: want [{'S' if prompt is None else prompt.shape[0]}, Lc, {self.D}]")
        S0 = prompt.shape[0] if prompt is not None else pre_c.shape[0]
        cfg_on = scaler.guidance_scale > 1
        passes = (3 if scaler.extra_pass else 2) if cfg_on else 1
        B = S0 // 2 if cfg_on else S0
        generator = inputs.get("generator", None)
        host_rng = generator is not None and generator.device.type == "cpu"
        rng_dev = "cpu" if host_rng else dev
        steps = inputs.get("num_diffusion_steps", 25)
        timesteps, coefs, ancestral = sampler_plan(m.sample_scheduler, steps)
        num_preds = [int(v) for v in inputs["num_preds"] if v > 0]
        latents = list(inputs.get("latents", []) or [])
        prefilled = bool(latents)  # first frame given (image-to-video): frame 0 is not generated (transformer_3d.py:159-160)
        gen_frames = [t for t in range(T) if not (t == 0 and prefilled)]
        if not gen_frames:
            return torch.stack([v.to(device=dev, dtype=dtype) for v in latents], dim=2)

        # ---- condition rows of every guidance pass: [cond ; uncond ; third] (guidance_scaler.py:46-57 expand_text)
        prompt = None if prompt is None else prompt.to(device=dev, dtype=dtype)
        motion = None
        if m.motion_embed is not None and inputs.get("motion_flow", None) is not None:  # transformer_3d.py:72-75
            flow, fps = inputs.get("motion_flow"), inputs.get("fps", None)
            rep = 2 if cfg_on else 1
            flow = list(flow) * rep if flow else [m.motion_embed.base_flow] * S0
            fps = list(fps) * rep if fps else [m.motion_embed.base_fps] * S0
            motion = torch.tensor([flow, fps], dtype=_F32).t().contiguous()  # [S0, 2]
        if passes == 3:
            third = slice(B, 2 * B) if scaler.image_guidance_scale else slice(0, B)
            prompt = None if prompt is None else torch.cat([prompt, prompt[third]])
            pre_c = None if pre_c is None else torch.cat([pre_c, pre_c[third]])
            motion = torch.cat([motion, motion[third]]) if motion is not None else None

        # ---- random draws for the WHOLE batch, in the reference's order (embeddings.py:265; transformer_3d.py:131).
        # Batch-sharded runs (sharding.py) pass batch_shard = (lo, hi, total): every rank then draws the
        # tensors of the GLOBAL batch from the same seed and keeps its rows, so sharded(seed) == unsharded(seed).
        g_lo, g_hi, g_B = inputs.get("batch_shard", None) or (0, B, B)
        if g_hi - g_lo != B or not (0 <= g_lo <= g_hi <= g_B):
            raise ValueError(f"batch_shard {(g_lo, g_hi, g_B)} does not describe this call's {B} samples")
        order = inputs.get("pred_order", None)  # test hook: inject the generation order [B, N]
        if order is None:
            u = torch.empty(g_B, N, 1, dtype=_F32, device=rng_dev).uniform_(generator=generator)
            order = u[g_lo:g_hi].argsort(dim=1)[..., 0]
        order = order.to(dev).contiguous()
        m.mask_embed.pred_ids = order.unsqueeze(-1)
        noise_fn = inputs.get("noise_fn", None)  # test hook: replay recorded per-step noise (index: running AR step)
        noise_buf = torch.empty(g_B, C, H, W, dtype=_F32, device=rng_dev)
        # The rows handed to the lanes must OWN their memory. With patch size 1 (point sets) the patchify below is a pure view, and with
        # a device generator `t` is a slice of `noise_buf`, which the next AR step's draw overwrites on the main stream while a lane
        # stream may still be waiting to read this step's rows (the host runs a whole encoder pass ahead of the device): the lanes
        # then denoise from another step's draw, timing-dependently - same distribution, but not the seed's sequence, and not the same
        # from run to run. `.contiguous()` of the strided view is the copy that cuts the alias (0.8 MB per AR step at batch 32).
        def to_rows(t):
            r = t.to(dev).reshape(B, C, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(B, N, P)
            if not r.is_contiguous():
                return r.contiguous()  # strided view (patch size 1): the copy
            return r.clone() if r.data_ptr() == t.data_ptr() else r  # a contiguous view of `t` itself (one channel): copy as well

        def draw(i):
            """Per-AR-step draws: x_T canvas [B,N,P] and, for an ancestral sampler, one gaussian canvas per step with t > 0."""
            if noise_fn is not None:
                nz = noise_fn(i).to(_F32)
            else:
                nz = noise_buf.normal_(generator=generator)[g_lo:g_hi]
            extra = None
            if ancestral:  # scheduling_ddpm.py:303-305
                extra = [to_rows(torch.randn(g_B, C, H, W, generator=generator, device=rng_dev, dtype=_F32)[g_lo:g_hi])
                         if coefs[j][5] != 0.0 else None for j in range(steps)]
            return to_rows(nz), extra

        # measured (MI355X): two lanes +6 % at batch 8 (d48w768 / 1024 points) and +3 % at batch 32 (d48w1024 / 2048 points)
        # once the encoder GEMMs are persistent (one lane's row kernels, decoder launches and epilogue store bursts fill
        # the other's memory-idle K loops); four lanes lose 10 % (quarter-size GEMMs).
        lanes = int(inputs.get("lanes", 0)) or _ENV_LANES or (2 if B >= 4 else 1)
        lanes = max(1, min(lanes, B))
        main = torch.cuda.current_stream()
        bounds = [(B * k // lanes, B * (k + 1) // lanes) for k in range(lanes)]
        first = latents[-1].to(device=dev, dtype=_F32) if prefilled else None
        runs = []
        for k, (lo, hi) in enumerate(bounds):
            pick = lambda t: torch.cat([t[q * B + lo:q * B + hi] for q in range(passes)]).contiguous()
            # a single lane stays on the caller's stream unless that is the legacy default stream, which cannot be
            # captured (nova_decoder_denoise replays its launch sequence as a hipGraph)
            stream = main if (lanes == 1 and main != torch.cuda.default_stream(dev)) else self._lane_stream(k)
            ctx = dict(k=k, lo=lo, hi=hi, prompt=None if prompt is None else pick(prompt), pre=None if pre_c is None else pick(pre_c),
                       S=passes * (hi - lo), motion=None if motion is None else pick(motion),
                       order=order[lo:hi].contiguous(), stream=stream, inbox=None, frames=[],
                       first=None if first is None else first[lo:hi])
            runs.append((ctx, self._lane(ctx, inputs, T=T, passes=passes, timesteps=timesteps, coefs=coefs, ancestral=ancestral,
                                         num_preds=num_preds, cfg_on=cfg_on)))

        def advance(ctx, gen):
            if ctx["stream"] is not main:
                ctx["stream"].wait_stream(main)
                with torch.cuda.stream(ctx["stream"]):
                    return next(gen, None)
            return next(gen, None)

        step_no = 0
        for t in range(T):
            for ctx, gen in runs:  # condition prefix (t = 0) + conditioning encoder of frame t
                advance(ctx, gen)
            if t not in gen_frames:
                continue
            for n in num_preds:
                nz, extra = draw(step_no)
                step_no += 1
                for ctx, gen in runs:
                    lo, hi = ctx["lo"], ctx["hi"]
                    ctx["inbox"] = (nz[lo:hi], None if extra is None else [None if e is None else e[lo:hi] for e in extra])
                    if ctx["stream"] is not main:  # tensors made on the main stream, consumed on the lane's stream
                        nz.record_stream(ctx["stream"])
                        [e.record_stream(ctx["stream"]) for e in (extra or []) if e is not None]
                    advance(ctx, gen)
        nf = len(runs[0][0]["frames"])
        canvas = torch.empty(nf, B, N, P, dtype=_F32, device=dev)
        mask = torch.empty(B, N, dtype=_F32, device=dev)
        for ctx, gen in runs:
            if ctx["stream"] is not main:
                main.wait_stream(ctx["stream"])
            for f in range(nf):
                canvas[f, ctx["lo"]:ctx["hi"]] = ctx["frames"][f]
            mask[ctx["lo"]:ctx["hi"]] = ctx["mask"]
        m.mask_embed.mask, m.mask_embed.pred_pos = mask.unsqueeze(-1).to(dtype), sum(num_preds)
        x = canvas.reshape(nf, B, h, w, p, p, C).permute(1, 6, 0, 2, 4, 3, 5).reshape(B, C, nf, H, W).to(dtype)
        if prefilled:  # the given frames stay in the output list ahead of the generated ones (transformer_3d.py:139,163)
            x = torch.cat([torch.stack([v.to(device=dev, dtype=dtype) for v in latents], dim=2), x], dim=2)
        return x

    def _lane_stream(self, k):
        pool = self.__dict__.setdefault("_streams", {})
        if (k, self.dev) not in pool:
            pool[(k, self.dev)] = torch.cuda.Stream(device=self.dev)
        return pool[(k, self.dev)]

    def _motion_tokens(self, values):
        """MotionEmbed.forward (embeddings.py:119-137): [S, 2] (flow, fps) -> two condition tokens per row [S, 2, D]."""
        me = self.model.motion_embed
        toks = []
        for col, (w1, b1, w2, b2) in enumerate(self.motion):
            ang = values[:, col].reshape(-1, 1).float() * me.freq_m.reshape(1, -1)
            feats = torch.cat([ang.sin(), ang.cos()], dim=-1).to(device=self.dev, dtype=self.dtype).contiguous()
            toks.append(self._gemm(self._gemm(feats, w1, b1, self.D, hip.ACT_SILU), w2, b2, self.D))
        return torch.stack(toks, dim=1)

    def _mixer(self, first, cur):
        """`video_encoder.mixer(states['*'], c)` (transformer_3d.py:156-158): AdaLayerNorm with eps=None, i.e.
        first * (1 + scale) + shift with (scale, shift) = proj(lora(SiLU(cur))) (normalization.py:33-36,41-46)."""
        mx = self.model.video_encoder.mixer
        pk = self.misc
        cache = pk.__dict__.get("mixer", None)
        if cache is None:
            if not isinstance(mx.norm, torch.nn.Identity):
                raise NotImplementedError("a normalising video mixer (eps != None) is not built on the HIP path")
            if isinstance(mx.lora, torch.nn.Identity):
                cache = pk.mixer = (None, 0, pk.w(mx.proj.weight, self.dtype), pk.f(mx.proj.bias))
            else:
                # low-rank pair proj(lora(.)) with nothing between the two: a rank that is not a multiple of the GEMM tile width is
                # padded with zero rows of `lora` and zero columns of `proj` - the padded products are exact zeros, the result is unchanged
                lw, pw_ = mx.lora.weight.detach(), mx.proj.weight.detach()
                rank = lw.shape[0]
                rank_p = -(-rank // 128) * 128
                if rank_p != rank:
                    lw = torch.cat([lw, lw.new_zeros(rank_p - rank, lw.shape[1])])
                    pw_ = torch.cat([pw_, pw_.new_zeros(pw_.shape[0], rank_p - rank)], dim=1)
                cache = pk.mixer = (pk.w(lw, self.dtype), rank_p, pk.w(pw_, self.dtype), pk.f(mx.proj.bias))
        lora, rank, pw, pb = cache
        act = torch.empty_like(cur)
        hip.call("nova_silu_add_rows", cur.data_ptr(), None, act.data_ptr(), cur.shape[0], self.D, self.code, hip.stream_ptr())
        if lora is not None:
            act = self._gemm(act, lora, None, rank)
        mod = self._gemm(act, pw, pb, 2 * self.D)
        out = torch.empty_like(first)
        hip.call("nova_modulate_rows", first.data_ptr(), mod.data_ptr(), out.data_ptr(), first.shape[0], self.D, self.code,
                 hip.stream_ptr())
        return out

    def _lane(self, ctx, inputs, T, passes, timesteps, coefs, ancestral, num_preds, cfg_on):
        """Generator: the generation loop for the samples [lo, hi) of the batch. Yields after the conditioning encoder
        of every frame and after every AR step (the caller alternates lanes and feeds this step's noise rows through
        ctx["inbox"])."""
        m = self.model
        dev, dtype, D = self.dev, self.dtype, self.D
        scaler = _Guidance(inputs)
        ie, ve = m.image_encoder, m.video_encoder
        C, (H, W), p = ie.image_dim, ie.image_size, ie.patch_embed.patch_size
        h, w = H // p, W // p
        pv = ve.patch_embed.patch_size
        hv, wv = H // pv, W // pv
        N, Nv, P, Pv = h * w, hv * wv, p * p * C, pv * pv * C
        prompt, pre = ctx["prompt"], ctx["pre"]
        S, Lt = ctx["S"], (0 if prompt is None else prompt.shape[1])
        Lc = 0 if pre is None else pre.shape[1]
        B = S // passes
        steps = len(timesteps)
        renorm = float(scaler.guidance_renorm)
        nmax = max(num_preds) if num_preds else 1
        Lp = Lc + Lt + (2 if ctx["motion"] is not None else 0)  # condition prefix: given rows, text tokens, flow and fps tokens
        L2 = Nv + N
        mod_bytes = steps * S * nmax * (3 * self.dec.depth + 2) * D * torch.empty((), dtype=dtype).element_size()
        mod_steps = steps if (steps > 1 and mod_bytes <= self.MOD_HOIST_BYTES and not inputs.get("per_step_adaln", False)) else 1
        ws = self._workspace(S, B, N, max(L2, Lp + Nv), nmax, ctx["k"], mod_steps)
        # fp8 delayed-scaling state is per call: every generation starts from the documented first-step guess (64 / 448) and adapts
        # from its own earlier AR steps, so a seeded call gives the same points whatever ran before it on this engine
        ws.pop("q8", None)
        code, st = self.code, hip.stream_ptr
        extra_kind = (1 if scaler.image_guidance_scale else 2) if passes == 3 else 0
        extra_scale = float(scaler.image_guidance_scale or scaler.spatiotemporal_guidance_scale) if passes == 3 else 0.0

        # ---- condition prefix: TextEmbed.forward (embeddings.py:203-206) [+ MotionEmbed tokens, transformer_3d.py:72-75]
        parts = [] if pre is None else [pre]
        if prompt is not None:
            pr = prompt.reshape(S * Lt, -1).contiguous()
            parts.append(self._norm_rows(self._gemm(pr, self.text[0], self.text[1], D), self.text[2:]).view(S, Lt, D))
        if ctx["motion"] is not None:
            parts.append(self._motion_tokens(ctx["motion"]))
        c_txt = (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)).reshape(S * Lp, D).contiguous()
        temb = ws["temb"].get(steps)
        if temb is None:
            temb = ws["temb"][steps] = torch.empty(steps, D, dtype=dtype, device=dev)
        temb.copy_(self.timestep_table(timesteps))

        # ---- positions / absolute position tables
        rope_i = rope_i0 = pos_img = inv_freq = inv_freq_v = img_pe = vpos = None
        hd = D // self.heads
        rotary = m.image_pos_embed is not None
        if rotary:
            vpos = m.video_pos_embed.get_pos(T)[0].to(device=dev, dtype=_F32).reshape(T, Nv, 3).contiguous()  # frame t: (t, h, w)
            pos_img = m.image_pos_embed.get_pos(1)[0].to(device=dev, dtype=_F32).contiguous()
            inv_freq = m.image_pos_embed.inv_freq().to(device=dev, dtype=_F32).contiguous()
            inv_freq_v = m.video_pos_embed.inv_freq().to(device=dev, dtype=_F32).contiguous()
            rope_i = hip.rope_table(pos_img, None, Nv, inv_freq, 1, hd)
            rope_i0 = rope_i[:, :Nv].contiguous()  # the first AR step's table (condition prefix only)
        else:  # abs-PE: tokens + time_embed[t] + sincos (transformer_3d.py:154, embeddings.py:103-115)
            vpe = m.video_pos_embed
            frame = (torch.arange(T, dtype=_F32) / (T / vpe.base_t)).view(-1, 1)
            ang = frame * vpe.freq_t.reshape(1, -1)
            feats = torch.cat([ang.sin(), ang.cos()], dim=-1).to(device=dev, dtype=dtype).contiguous()
            w1, b1, w2, b2, g, bt = self.vtime
            t_emb = self._norm_rows(self._gemm(self._gemm(feats, w1, b1, D, hip.ACT_SILU), w2, b2, D), (g, bt))  # [T, D]
            v_pe = vpe.get_space_embed(dev, dtype)  # [Nv, D]
            img_pe = ie.pos_embed.get_space_embed(dev, dtype).contiguous()
        bos = m.mask_embed.bos_token.detach().to(dtype)
        ar = torch.arange(S, device=dev, dtype=torch.int32)
        cache = cap = None
        if T > 1:  # KV cache of the conditioning encoder (vision_transformer.py:55-60,125-126): prefix + T frames of Nv tokens
            cap = Lp + T * Nv
            cache = torch.empty(len(self.video.arr), S, cap, 2 * D, dtype=dtype, device=dev)
        mixing = T > 1 and not isinstance(ve.mixer, torch.nn.Identity)
        c_first = None
        canvas = ctx["first"]
        if canvas is not None:  # prefilled first frame [b, C, H, W] -> patch rows [b, N, P]
            canvas = canvas.reshape(B, C, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(B, N, P).contiguous()
        order = ctx["order"]

        for t in range(T):
            # ---- conditioning ViT of frame t (transformer_3d.py:150-158)
            if t == 0:
                vtok = bos.expand(Nv, D)
                if not rotary:
                    vtok = vtok + t_emb[0] + v_pe
                vtok = vtok.contiguous()
                Lq, pad = Lp + Nv, Lp
                xv = ws["x1"][: S * Lq]
                self._sequence(xv, c_txt, Lp, vtok, 0, None, S, B, Lp, Nv)
            else:  # patch embedding (patch 2p) of the previous frame
                img = canvas.reshape(B, h, w, p, p, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)
                rows = img.reshape(B, C, hv, pv, wv, pv).permute(0, 2, 4, 3, 5, 1).reshape(B, Nv, Pv).contiguous()
                Lq, pad = Nv, 0
                xv = ws["x1"][: S * Lq]
                hip.call("nova_patch_embed_rows", rows.data_ptr(), self.vpatch[0], self.vpatch[1], xv.data_ptr(), S, B, Nv, Pv, D, code, st())
                if not rotary:
                    xv.view(S, Nv, D).add_(t_emb[t] + v_pe)
            if extra_kind == 1:  # image guidance: the uncond pass sees bare bos tokens (expand(c, padding=bos), guidance_scaler.py:42-43)
                xv.view(S, Lq, D)[B:2 * B, pad:] = bos
            rope_v = hip.rope_table(vpos[t].contiguous(), None, pad, inv_freq_v, 1, hd) if rotary else None
            if cache is None:
                self._blocks(self.video, xv, S, Lq, rope_v, 1, ws)
            else:
                hip.call("nova_vit_blocks_forward_kv", self.video.arr, len(self.video.arr), xv.data_ptr(), S, Lq, D, self.heads,
                         self.hidden, hip.ptr(rope_v), 1, cache.data_ptr(), cap, 0 if t == 0 else Lp + t * Nv,
                         ws["qkv"].data_ptr(), ws["a"].data_ptr(), ws["b"].data_ptr(), ws["h"].data_ptr(), code, st())
            vrows = (ar[:, None] * Lq + pad + torch.arange(Nv, device=dev, dtype=torch.int32)[None]).reshape(-1).contiguous()
            c = self._norm_rows(xv, self.vnorm, gather=vrows)  # [S*Nv, D]
            if mixing:
                c_first = c if t == 0 else c_first
                c = self._mixer(c_first, c) if t else c
            if t == 0 and canvas is not None:  # prefilled first frame: nothing to generate
                yield "prefix"
                continue

            # ---- masked autoregressive loop of frame t (transformer_3d.py:115-133)
            canvas = torch.zeros(B, N, P, dtype=_F32, device=dev)
            mask = torch.ones(B, N, dtype=_F32, device=dev)
            ctx["mask"] = mask
            yield "prefix"
            done = 0
            for i, n in enumerate(num_preds):
                scaler.decay_guidance_scale((i + 1) / len(num_preds))
                if cfg_on and scaler.guidance_scale <= 1:
                    raise NotImplementedError("guidance decaying to <= 1 inside a CFG run is undefined in the reference")
                plan = (hip.SamplerStep * steps)()
                for j, tt in enumerate(timesteps):
                    g = 1.0 if (cfg_on and scaler.guidance_trunc and float(tt) < scaler.guidance_trunc) else float(scaler.guidance_scale)
                    plan[j] = hip.SamplerStep(g if cfg_on else 1.0, *coefs[j], extra_scale, extra_kind)
                z0 = ws["z0"]
                hip.call("nova_embed_canvas", canvas.data_ptr(), mask.data_ptr(), self.patch[0], self.patch[1], self.mask_token,
                         hip.ptr(img_pe), z0.data_ptr(), B, N, P, D, code, st())
                prev_ids = ws["ids_prev"][: B * done].view(B, done)
                prev_ids.copy_(order[:, :done])
                pred_ids = ws["ids_pred"][: B * n].view(B, n)
                pred_ids.copy_(order[:, done : done + n])
                mask.scatter_(1, pred_ids, 0.0)
                # first half: [c ; known tokens in generation order]
                L1 = Nv + done
                x1 = ws["x1"][: S * L1]
                self._sequence(x1, c, Nv, z0, N, prev_ids if done else None, S, B, Nv, done)
                rope1 = hip.rope_table(pos_img, prev_ids, Nv, inv_freq, B, hd, out=ws["rope1"]) if (rotary and done) else (
                    rope_i0 if rotary else None)
                self._blocks(self.enc1, x1, S, L1, rope1, B if (rotary and done) else 1, ws)
                # second half: [c' ; full canvas with the known tokens scattered back]
                x2 = ws["x2"][: S * L2]
                self._sequence(x2, x1, L1, z0, N, None, S, B, Nv, N)
                if done:
                    hip.call("nova_scatter_tokens", x1.data_ptr(), prev_ids.data_ptr(), x2.data_ptr(), S, B, Nv, N, done, D, code, st())
                if self.enc2_head is not None:
                    self._blocks(self.enc2_head, x2, S, L2, rope_i, 1, ws)
                y = self._last_block_rows(x2, S, B, L2, Nv, n, pred_ids, pos_img, inv_freq, rope_i, hd, ws)
                # final LN only on the rows predicted now, then the condition projection (time term added per step)
                zc = self._norm_rows(y, self.inorm, out=ws["lz"][: S * n])
                w1, b1, w2, b2 = self.dec.time[1]
                zc = self._gemm(self._gemm(zc, w1, b1, D, hip.ACT_SILU, out=ws["lz2"][: S * n]), w2, b2, D, out=ws["dz"][: S * n])
                # this step's noise rows (drawn by the caller for the whole batch)
                nz, extra = ctx["inbox"]
                idx = pred_ids[..., None].expand(-1, -1, P)
                x_n = torch.gather(nz, 1, idx, out=ws["dx"][: B * n * P].view(B, n, P))
                step_noise = echo = ws_v = None
                if ancestral:
                    step_noise = torch.stack([torch.zeros(B, n, P, dtype=_F32, device=dev) if e is None else e.gather(1, idx)
                                              for e in extra]).contiguous()
                if renorm < 1 and cfg_on and ancestral:
                    # the rows that only echo x_t in the reference (guidance_scaler.py:67-72 takes its norms over all N rows) are carried
                    # explicitly: the ancestral step adds noise to them too (scheduling_ddpm.py:303-312), so no scalar stands in for them
                    rest = torch.ones(B, N, dtype=torch.bool, device=dev).scatter_(1, pred_ids.long(), False)
                    ridx = rest.nonzero()[:, 1].view(B, N - n, 1).expand(-1, -1, P)
                    echo_rows = nz.gather(1, ridx).contiguous()
                    echo_noise = torch.stack([torch.zeros(B, N - n, P, dtype=_F32, device=dev) if e is None else e.gather(1, ridx)
                                              for e in extra]).contiguous()
                    ws_v = torch.empty(3 * B * n * P, dtype=_F32, device=dev)
                    hip.call("nova_decoder_denoise_echo", ctypes.byref(self.dec.struct), zc.data_ptr(), temb.data_ptr(), x_n.data_ptr(), plan,
                             hip.ptr(step_noise), renorm, echo_rows.data_ptr(), echo_noise.data_ptr(), N - n, steps, S, B, n, P, D,
                             ws["da"].data_ptr(), ws["du"].data_ptr(), ws["dh"].data_ptr(), ws["df"].data_ptr(), ws["dg"].data_ptr(),
                             ws["dmod"].data_ptr(), ws_v.data_ptr(), mod_steps, code, st())
                    canvas.scatter_(1, idx, x_n)
                    done += n
                    if i == len(num_preds) - 1:
                        ctx["frames"].append(canvas)
                    yield i
                    continue
                if renorm < 1 and cfg_on:  # squared norm of the rows that only echo x_t in the reference (guidance_scaler.py:67-72)
                    # per-sample sums in float64: an f32 reduction's summation order follows the tensor's batch size (torch tiles it by shape),
                    # which would make a sample's result depend - at rounding level - on how many samples share its lane
                    echo = (nz.double().pow(2).sum((1, 2)) - x_n.double().pow(2).sum((1, 2))).clamp_min(0).float().contiguous()
                    ws_v = torch.empty(3 * B * n * P, dtype=_F32, device=dev)
                hip.call("nova_decoder_denoise", ctypes.byref(self.dec.struct), zc.data_ptr(), temb.data_ptr(), x_n.data_ptr(), plan,
                         hip.ptr(step_noise), renorm if cfg_on else 1.0, hip.ptr(echo), steps, S, B, n, P, D, ws["da"].data_ptr(),
                         ws["du"].data_ptr(), ws["dh"].data_ptr(), ws["df"].data_ptr(), ws["dg"].data_ptr(), ws["dmod"].data_ptr(),
                         hip.ptr(ws_v), mod_steps, code, st())
                canvas.scatter_(1, idx, x_n)
                done += n
                if i == len(num_preds) - 1:
                    ctx["frames"].append(canvas)
                yield i


# --------------------------------------------------------------------------------------------
# module-level entry points (what `Block.forward`, `VisionTransformer.forward` and
# `DiffusionMLP.forward` of the drop-in package call for device tensors without autograd)
# --------------------------------------------------------------------------------------------
def _cached_pack(owner, key, build):
    sig = (key, _params_signature(owner))
    cache = owner.__dict__.setdefault("_nova_packs", {})
    if cache.get("sig") != sig:
        cache.clear()
        cache.update(sig=sig, pack=build())
    return cache["pack"]


def block_stack_forward(blocks, x, pe_func=None):
    """Run a list of `Block` modules on x [S, L, D] through nova_vit_blocks_forward (returns a new tensor)."""
    hip.load()
    S, L, D = x.shape
    first = blocks[0]
    pack = _cached_pack(first, ("vit", len(blocks), tuple(id(b) for b in blocks), x.dtype),
                        lambda: pack_vit_blocks(blocks, x.dtype))
    heads, hidden = first.attn.num_heads, first.mlp.fc1.out_features
    rope = None
    if pe_func is not None:
        w = pe_func.weight[:, 0]
        rope = torch.stack([w[..., 0, 0], w[..., 1, 0]], dim=-1).float().contiguous()
    rows = S * L
    out = x.reshape(rows, D).clone()
    e = lambda n: torch.empty(rows, n, dtype=x.dtype, device=x.device)
    qkv, a, b, h = e(3 * D), e(D), e(D), e(hidden)
    hip.call("nova_vit_blocks_forward", pack.arr, len(pack.arr), out.data_ptr(), S, L, D, heads, hidden, hip.ptr(rope),
             rope.shape[0] if rope is not None else 1, qkv.data_ptr(), a.data_ptr(), b.data_ptr(), h.data_ptr(),
             hip.dtype_code(x.dtype), hip.stream_ptr())
    return out.view(S, L, D)


def decoder_forward(dec, x_img, timestep, z, pred_ids=None):
    """`DiffusionMLP.forward` (diffusion_mlp.py:89-99) for device tensors: one velocity prediction."""
    hip.load()
    dtype, dev = z.dtype, z.device
    code, st = hip.dtype_code(dtype), hip.stream_ptr
    pk = _cached_pack(dec, ("dec", dtype), lambda: pack_decoder(dec, dtype))
    pe = dec.patch_embed
    S, D = z.shape[0], z.shape[-1]
    if x_img.dim() != 4:
        raise NotImplementedError("decoder_forward expects the noisy canvas as [S, C, H, W]")
    pe.height, pe.width = x_img.size(-2) // pe.patch_size, x_img.size(-1) // pe.patch_size
    echo = pe.patchify(x_img)
    N, P = echo.shape[1], echo.shape[2]
    ids = pred_ids[..., 0] if pred_ids is not None else torch.arange(N, device=dev).expand(S, -1)
    n = ids.shape[1]
    x_rows = echo.float().gather(1, ids[..., None].expand(-1, -1, P)).contiguous()  # [S, n, P] f32
    rows = (torch.arange(S, device=dev)[:, None] * z.shape[1] + ids).to(torch.int32).reshape(-1).contiguous()
    zf = z.reshape(-1, D).contiguous()
    zg = torch.empty(S * n, D, dtype=dtype, device=dev)
    hip.call("nova_build_sequence", None, 0, zf.data_ptr(), z.shape[1], ids.contiguous().data_ptr(), zg.data_ptr(), S, S, 0, n, D,
             code, st())
    gemm = lambda a, w, b, N_, act=hip.ACT_NONE: _gemm_ptr(a, w, b, N_, act, code)
    w1, b1, w2, b2 = pk.time[1]
    cond = gemm(gemm(zg, w1, b1, D, hip.ACT_SILU), w2, b2, D)
    # timestep features -> timestep_proj, one row per sequence
    t = torch.as_tensor(timestep, device=dev).float().reshape(-1).expand(S).contiguous()
    freq = torch.arange(128, dtype=_F32, device=dev).mul(-9.210340371976184 / 128).exp()
    feats = torch.empty(S, 256, dtype=dtype, device=dev)
    hip.call("nova_timestep_freq", t.data_ptr(), freq.data_ptr(), feats.data_ptr(), S, 256, code, st())
    w1, b1, w2, b2 = pk.time[0]
    temb = gemm(gemm(feats, w1, b1, D, hip.ACT_SILU), w2, b2, D)
    es = zg.element_size()
    act = torch.empty_like(cond)
    for s in range(S):  # SiLU(cond + temb[s]) per sequence (timesteps may differ per sequence)
        hip.call("nova_silu_add_rows", cond.data_ptr() + s * n * D * es, temb.data_ptr() + s * D * es,
                 act.data_ptr() + s * n * D * es, n, D, code, st())
    depth = pk.depth
    mod = gemm(act, pk.struct.adaln_w, pk.struct.adaln_b, (3 * depth + 2) * D)
    u = torch.empty(S * n, D, dtype=dtype, device=dev)
    hip.call("nova_patch_embed_rows", x_rows.data_ptr(), pk.struct.patch_w, pk.struct.patch_b, u.data_ptr(), S, S, n, P, D,
             code, st())
    for i in range(depth):
        blk = pk.blocks[i]
        h = hip.row_norm(u, mod=mod, scale_off=i * 3 * D, shift_off=i * 3 * D + D, eps=1e-6)
        g = gemm(gemm(h, blk.fc1_w, blk.fc1_b, D, hip.ACT_SILU), blk.fc2_w, blk.fc2_b, D)
        hip.call("nova_row_norm", g.data_ptr(), u.data_ptr(), blk.norm2_w, blk.norm2_b, mod.data_ptr(), mod.shape[1], -1, -1,
                 i * 3 * D + 2 * D, u.data_ptr(), None, S * n, D, 1e-5, code, st())
    h = hip.row_norm(u, mod=mod, scale_off=depth * 3 * D, shift_off=depth * 3 * D + D, eps=1e-6)
    pred = torch.zeros(S, n, P, dtype=_F32, device=dev)
    hip.call("nova_head_cfg_euler", h.data_ptr(), pk.struct.head_w, pk.struct.head_b, pred.data_ptr(), S, n, P, D, 1.0, 0, 1.0,
             code, st())
    pred = pred.to(dtype)
    return pred if pred_ids is None else echo.scatter(1, ids[..., None].expand(-1, -1, P), pred)


def _gemm_ptr(a, w_ptr, b_ptr, N, act, code):
    M, K = a.shape
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    hip.call("nova_gemm_bias_act", a.data_ptr(), w_ptr, b_ptr, out.data_ptr(), M, N, K, act, code, hip.stream_ptr())
    return out
