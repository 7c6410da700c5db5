"""ctypes binding of libnova_hip.so (C ABI in include/nova_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or the
current device is not gfx950, every call raises ``NovaHipError``. PyTorch is used here only for
device memory (``tensor.data_ptr()``) and the current HIP stream.
"""

import ctypes
import os
import threading

import torch

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_GELU_ERF, ACT_SILU = 0, 1, 2

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnova_hip.so")
_lock = threading.Lock()
_lib = None
_device_ok = False

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float


class NovaHipError(RuntimeError):
    """Raised when libnova_hip.so is missing, mis-built, or a call into it fails."""


class VitBlock(ctypes.Structure):
    """``nova_vit_block`` (include/nova_hip.h)."""

    _fields_ = [(k, c_void_p) for k in (
        "qkv_w", "qkv_b", "proj_w", "proj_b", "norm1_w", "norm1_b",
        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm2_w", "norm2_b")]


class VitBlockFp8(ctypes.Structure):
    """``nova_vit_block_fp8`` (include/nova_hip.h)."""

    _fields_ = [(k, c_void_p) for k in ("qkv_w8", "qkv_ws", "fc1_w8", "fc1_ws", "fc2_w8", "fc2_ws")]


class MlpBlock(ctypes.Structure):
    """``nova_mlp_block`` (include/nova_hip.h)."""

    _fields_ = [(k, c_void_p) for k in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm2_w", "norm2_b")]


class Decoder(ctypes.Structure):
    """``nova_decoder`` (include/nova_hip.h)."""

    _fields_ = [("depth", c_int), ("blocks", ctypes.POINTER(MlpBlock))] + [(k, c_void_p) for k in (
        "adaln_w", "adaln_b", "patch_w", "patch_b", "head_w", "head_b")]


class SamplerStep(ctypes.Structure):
    """``nova_sampler_step`` (include/nova_hip.h)."""

    _fields_ = [(k, c_float) for k in ("guidance", "kx", "kv", "clip", "c0", "cx", "sigma", "extra_scale")] + [("extra_kind", c_int)]


# name -> argtypes; every function returns int status. Must list EVERY symbol of nova_hip.h
# (tests/test_abi.py checks the header against this table and against the built library).
SIGNATURES = {
    "nova_gemm_bias_act": [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    "nova_qkv_rope": [c_void_p] * 5 + [c_int] * 6 + [c_void_p],
    "nova_quantize_rows_fp8": [c_void_p] * 3 + [ctypes.c_longlong, c_int, c_void_p],
    "nova_gemm_fp8_bias_act": [c_void_p] * 6 + [c_int] * 4 + [c_void_p],
    "nova_qkv_rope_cols": [c_void_p] * 5 + [c_int] * 8 + [c_void_p],
    "nova_rope_table": [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_void_p],
    "nova_attn_fwd": [c_void_p] * 4 + [c_int] * 5 + [c_long] * 3 + [c_float, c_int, c_void_p],
    "nova_row_norm": [c_void_p] * 5 + [c_long] + [c_int] * 3 + [c_void_p, c_void_p, c_long, c_int, c_float, c_int, c_void_p],
    "nova_embed_canvas": [c_void_p] * 7 + [c_int] * 5 + [c_void_p],
    "nova_build_sequence": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p],
    "nova_scatter_tokens": [c_void_p] * 3 + [c_int] * 7 + [c_void_p],
    "nova_silu_add_rows": [c_void_p] * 3 + [c_long, c_int, c_int, c_void_p],
    "nova_timestep_freq": [c_void_p] * 3 + [c_int] * 3 + [c_void_p],
    "nova_patch_embed_rows": [c_void_p] * 4 + [c_int] * 6 + [c_void_p],
    "nova_head_cfg_euler": [c_void_p] * 4 + [c_int] * 4 + [c_float, c_int, c_float, c_int, c_void_p],
    "nova_vit_blocks_forward": [ctypes.POINTER(VitBlock), c_int, c_void_p] + [c_int] * 5 + [c_void_p, c_int]
    + [c_void_p] * 4 + [c_int, c_void_p],
    "nova_row_norm_fp8": [c_void_p] * 7 + [c_long, c_int, c_float, c_void_p],
    "nova_gemm_fp8_gelu_q8": [c_void_p] * 6 + [c_int] * 3 + [c_void_p, c_void_p, c_void_p],
    "nova_qkv_rope_fp8": [c_void_p] * 7 + [c_int] * 5 + [c_float, c_void_p],
    "nova_vit_blocks_forward_fp8": [ctypes.POINTER(VitBlock), ctypes.POINTER(VitBlockFp8), c_int, c_void_p] + [c_int] * 5
    + [c_void_p, c_int] + [c_void_p] * 10 + [c_void_p],
    "nova_vit_blocks_forward_kv": [ctypes.POINTER(VitBlock), c_int, c_void_p] + [c_int] * 5 + [c_void_p, c_int]
    + [c_void_p, c_long, c_long] + [c_void_p] * 4 + [c_int, c_void_p],
    "nova_pointset_nn_dist": [c_void_p] * 3 + [c_int] * 3 + [c_float, c_float, c_int, c_void_p],
    "nova_pointset_pairwise_dist": [c_void_p] * 3 + [c_int] * 3 + [c_float, c_float, c_void_p],
    "nova_modulate_rows": [c_void_p] * 3 + [c_long, c_int, c_int, c_void_p],
    "nova_attn_fwd_lse": [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_long, c_long, c_void_p, c_void_p],
    "nova_attn_bwd": [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_float, c_void_p, c_void_p],
    "nova_row_norm_bwd": [c_void_p] * 5 + [c_long, c_int, c_int, c_int] + [c_void_p] * 4 + [c_int, c_long, c_int, c_float, c_int, c_void_p],
    "nova_act_fwd": [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p],
    "nova_act_bwd": [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p],
    "nova_row_norm_chain": [c_void_p] * 5 + [c_long, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "nova_adaln_fc1": [c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_long]
    + [c_int] * 4 + [c_void_p],
    "nova_decoder_denoise": [ctypes.POINTER(Decoder), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]
    + [c_int] * 6 + [c_void_p] * 7 + [c_int, c_int, c_void_p],
    "nova_decoder_denoise_echo": [ctypes.POINTER(Decoder), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]
    + [c_int] * 7 + [c_void_p] * 7 + [c_int, c_int, c_void_p],
}
SIGNATURES["nova_prof_enable"] = [c_int]
SIGNATURES["nova_debug_force_gemm_tile"] = [c_int]
SIGNATURES["nova_debug_set_graphs"] = [c_int]
SIGNATURES["nova_debug_set_attn_variant"] = [c_int]
SIGNATURES["nova_debug_drop_graphs"] = []
SIGNATURES["nova_debug_graph_stats"] = [ctypes.POINTER(c_long), ctypes.POINTER(c_long)]
SIGNATURES["nova_prof_collect"] = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(ctypes.c_longlong), c_int]
PROF_SLOTS = ["gemm_bias", "gemm_bias_gelu", "gemm_bias_silu", "qkv_gemm_rope", "attention", "row_norm", "gemm_small_tile", "gemm_bias_wide_k",
              "token_plumbing", "decoder_glue"]
PLAIN = {"nova_version": (c_int, []), "nova_last_error": (ctypes.c_char_p, []), "nova_check_device": (c_int, [])}


def lib_path() -> str:
    return _LIB_PATH


ABI_VERSION = 401  # == NOVA_HIP_VERSION of include/nova_hip.h (tests/test_abi.py compares the two)


def load(check_device=True):
    """Load (once) and return the ctypes handle. Raises NovaHipError if it cannot."""
    global _lib, _device_ok
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise NovaHipError(
                    f"{_LIB_PATH} not found: the HIP extension is not built. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C nova_pointcloud_amd/csrc`). "
                    "There is no CPU fallback for the generation path."
                )
            try:
                lib = ctypes.CDLL(_LIB_PATH)
            except OSError as e:  # pragma: no cover
                raise NovaHipError(f"cannot load {_LIB_PATH}: {e}") from e
            lib.nova_version.restype = c_int
            if lib.nova_version() != ABI_VERSION:  # a stale build would take e.g. `stream` where `key_limit` now sits
                raise NovaHipError(f"{_LIB_PATH} is C ABI version {lib.nova_version()}, this package binds version {ABI_VERSION} "
                                   "(include/nova_hip.h NOVA_HIP_VERSION): rebuild it with `make -C nova_pointcloud_amd/csrc`")
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes, fn.restype = argtypes, c_int
            for name, (res, argtypes) in PLAIN.items():
                fn = getattr(lib, name)
                fn.argtypes, fn.restype = argtypes, res
            _lib = lib
        if check_device and not _device_ok:
            if not torch.cuda.is_available():
                raise NovaHipError("no GPU visible: libnova_hip needs an MI355X (gfx950) device")
            rc = _lib.nova_check_device()
            if rc != 0:
                raise NovaHipError(_lib.nova_last_error().decode())
            _device_ok = True
    return _lib


def call(name, *args):
    """Invoke a status-returning entry point; raise NovaHipError with the library's message on failure."""
    lib = _lib if (_lib is not None and _device_ok) else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NovaHipError(f"{name} failed ({rc}): {lib.nova_last_error().decode()}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise NovaHipError(f"libnova_hip supports float32, bfloat16 and float16 activations, got {dtype}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NovaHipError("libnova_hip got a CPU tensor")
    if not t.is_contiguous():
        raise NovaHipError("libnova_hip needs contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise NovaHipError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


# --------------------------------------------------------------------------------------------
# thin tensor-level wrappers (used by the diffnext modules, the engine and the GPU tests)
# --------------------------------------------------------------------------------------------
def gemm_bias_act(a, w, bias=None, act=ACT_NONE, out=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias)."""
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == a.dtype
    out = a.new_empty(M, N) if out is None else out
    call("nova_gemm_bias_act", ptr(a), ptr(w), ptr(bias, torch.float32), ptr(out), M, N, K, act,
         dtype_code(a.dtype), stream_ptr())
    return out


def qkv_rope(x, w, bias, rope, S, L, heads, out=None):
    """Fused QKV projection (+ RoPE when ``rope`` [nb, L, hd/2, 2] f32 is given)."""
    D = x.shape[-1]
    out = x.new_empty(S * L, 3 * D) if out is None else out
    nb = rope.shape[0] if rope is not None else 1
    if rope is not None:
        assert rope.shape[1] == L and rope.shape[2] * 2 == D // heads
    call("nova_qkv_rope", ptr(x), ptr(w), ptr(bias, torch.float32), ptr(rope, torch.float32), ptr(out),
         S, L, D, heads, nb, dtype_code(x.dtype), stream_ptr())
    return out


def quantize_rows_fp8(x):
    """Per-row dynamic quantisation of bf16 rows [rows, D] to OCP e4m3 bytes + f32 row scales (amax / 448)."""
    rows, D = x.shape
    q = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
    scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("nova_quantize_rows_fp8", ptr(x, torch.bfloat16), ptr(q), ptr(scale), rows, D, stream_ptr())
    return q, scale


def gemm_fp8_bias_act(a8, a_scale, w8, w_scale, bias, act=0, out=None):
    """bf16 out[M,N] = act((a8 . w8^T) * a_scale[m] * w_scale[n] + bias); a8 [M,K], w8 [N,K] e4m3 bytes (uint8 views)."""
    M, K = a8.shape
    N = w8.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device) if out is None else out
    call("nova_gemm_fp8_bias_act", ptr(a8), ptr(a_scale, torch.float32), ptr(w8), ptr(w_scale, torch.float32),
         ptr(bias, torch.float32), ptr(out), M, N, K, act, stream_ptr())
    return out


def rope_table(pos, ids, pad, inv_freq, nb, hd, out=None):
    """cos/sin table [nb, pad + n_tok, hd/2, 2] f32. pos [n_pos, 3] f32, ids [nb, n_tok] int64 or None.
    `out`: a flat f32 buffer of at least that many elements (the hot loop's workspace slot); a view of it is returned."""
    n_pos = pos.shape[0]
    n_tok = ids.shape[1] if ids is not None else n_pos
    if out is None:
        out = torch.empty(nb, pad + n_tok, hd // 2, 2, dtype=torch.float32, device=pos.device)
    else:
        out = out[: nb * (pad + n_tok) * hd].view(nb, pad + n_tok, hd // 2, 2)
    call("nova_rope_table", ptr(pos, torch.float32), ptr(ids, torch.int64), ptr(out), nb, pad, n_tok, n_pos, hd,
         ptr(inv_freq, torch.float32), stream_ptr())
    return out


def attn_fwd_packed(qkv, S, L, heads, out=None):
    """Attention reading q/k/v in place from the fused [S*L, 3D] buffer; returns merged heads [S*L, D]."""
    D = qkv.shape[-1] // 3
    hd = D // heads
    out = qkv.new_empty(S * L, D) if out is None else out
    es = qkv.element_size()
    base = ptr(qkv)
    call("nova_attn_fwd", base, base + D * es, base + 2 * D * es, ptr(out), S, heads, L, L, hd, 3 * D, 3 * D, D,
         float(hd) ** -0.5, dtype_code(qkv.dtype), stream_ptr())
    return out


def row_norm(x, out=None, gamma=None, beta=None, mod=None, scale_off=-1, shift_off=-1, gate_off=-1, res=None,
             gather=None, eps=1e-5, rows=None):
    D = x.shape[-1]
    rows = (gather.numel() if gather is not None else x.numel() // D) if rows is None else rows
    out = x.new_empty(rows, D) if out is None else out
    mod_ld = mod.shape[-1] if mod is not None else 0
    call("nova_row_norm", ptr(x), ptr(out), ptr(gamma, torch.float32), ptr(beta, torch.float32), ptr(mod), mod_ld,
         scale_off, shift_off, gate_off, ptr(res), ptr(gather, torch.int32), rows, D, float(eps),
         dtype_code(x.dtype), stream_ptr())
    return out


def adaln_fc1(x, mod, scale_off, shift_off, w, bias=None, act=ACT_SILU, eps=1e-6, out=None):
    """act((LN(x) * (1 + mod[:, scale_off:+D]) + mod[:, shift_off:+D]) @ w^T + bias): DiffusionBlock's modulate -> fc1 -> SiLU
    (diffusion_mlp.py:41-47); one launch at small row counts for bf16 rows of width 768 / 1024, else LN launch + GEMM."""
    rows, D = x.shape
    N = w.shape[0]
    out = x.new_empty(rows, N) if out is None else out
    h = x.new_empty(rows, D)
    call("nova_adaln_fc1", ptr(x), ptr(mod), mod.shape[-1], scale_off, shift_off, float(eps), ptr(w), ptr(bias, torch.float32),
         ptr(h), ptr(out), rows, N, D, act, dtype_code(x.dtype), stream_ptr())
    return out


def set_graphs(on=True):
    """hipGraph replay of nova_decoder_denoise's launch sequence (default on; NOVA_GRAPHS=0 in the environment disables)."""
    call("nova_debug_set_graphs", 1 if on else 0)


def graph_stats():
    """(graphs captured, graphs replayed) by the calling thread."""
    c, r = c_long(0), c_long(0)
    call("nova_debug_graph_stats", ctypes.byref(c), ctypes.byref(r))
    return c.value, r.value


def prof_enable(on=True):
    call("nova_prof_enable", 1 if on else 0)


def prof_collect():
    """{slot name: (milliseconds, algorithmic work, launches)} since the last collect; waits for the events."""
    n = len(PROF_SLOTS)
    ms, work, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_longlong * n)()
    call("nova_prof_collect", ms, work, cnt, n)
    return {PROF_SLOTS[i]: (ms[i], work[i], cnt[i]) for i in range(n)}
