"""ctypes binding of ``csrc/libpdengine.so`` (C ABI: ``include/pdengine.h``).

There is no CPU fallback: if the HIP library is missing, or no MI355X is visible,
construction raises.  Arrays cross the boundary as float32 in the reference's layouts
(NCHW latents/images, [B, L, D] context); NumPy arrays are passed as host pointers,
torch CUDA tensors as device pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .weights import ModelConfig, SD15

PD_PREC_BF16, PD_PREC_F32, PD_PREC_F16, PD_PREC_F16X2 = 0, 1, 2, 3
PRECISIONS = {"bf16": PD_PREC_BF16, "f32": PD_PREC_F32, "fp32": PD_PREC_F32, "f16": PD_PREC_F16, "fp16": PD_PREC_F16,
              "f16x2": PD_PREC_F16X2}
PD_MEM_HOST, PD_MEM_DEVICE = 0, 1
PD_DT_F32, PD_DT_F16, PD_DT_BF16 = 0, 1, 2
PD_GET_LATENTS, PD_GET_PRED_X0, PD_GET_EPS = 0, 1, 2
PD_MAX_LEVELS = 8
PD_NUM_CONTROL = 13
PD_COMM_ID_BYTES = 128

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libpdengine.so")


class PdError(RuntimeError):
    pass


class pd_config(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("hint_channels", C.c_int32),
        ("query_channels", C.c_int32), ("model_channels", C.c_int32), ("num_levels", C.c_int32),
        ("channel_mult", C.c_int32 * PD_MAX_LEVELS), ("num_res_blocks", C.c_int32),
        ("num_attn_res", C.c_int32), ("attention_resolutions", C.c_int32 * PD_MAX_LEVELS),
        ("num_heads", C.c_int32), ("context_dim", C.c_int32), ("context_len", C.c_int32),
        ("hint_widths", C.c_int32 * 7), ("timesteps", C.c_int32), ("linear_start", C.c_double),
        ("linear_end", C.c_double), ("precision", C.c_int32), ("stream_f32", C.c_int32),
        ("vae_ch", C.c_int32), ("vae_num_levels", C.c_int32), ("vae_ch_mult", C.c_int32 * PD_MAX_LEVELS),
        ("vae_num_res_blocks", C.c_int32), ("vae_out_ch", C.c_int32), ("scale_factor", C.c_double),
        ("text_vocab", C.c_int32), ("text_layers", C.c_int32), ("text_heads", C.c_int32), ("text_ff", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class pd_sample_args(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("steps", C.c_int32),
        ("eta", C.c_float), ("cfg_scale", C.c_float), ("use_cfg", C.c_int32), ("guess_mode", C.c_int32),
        ("only_mid_control", C.c_int32), ("temperature", C.c_float), ("mem", C.c_int32),
        ("x_T", C.c_void_p), ("ctx_cond", C.c_void_p), ("ctx_uncond", C.c_void_p), ("pair", C.c_void_p),
        ("query", C.c_void_p), ("pair_uncond", C.c_void_p), ("query_uncond", C.c_void_p),
        ("control_scales", C.c_void_p), ("control_scales_step", C.c_void_p), ("noise", C.c_void_p),
        ("timesteps", C.c_void_p), ("reserved", C.c_int32 * 6),
    ]


_lib = None


def _torch_runtime_first():
    """PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so) under the same soname as the system one that
    libpdengine.so links.  The dynamic loader keeps whichever is loaded first for the whole process: with the system
    runtime first, torch (built against its own) later reports "No HIP GPUs are available"; with torch's first, both
    work.  So when torch is installed, load it -- and bring its device runtime up -- before libpdengine.so is opened.
    CUDA tensors can then be handed to the engine as device pointers whatever the caller's import order."""
    try:
        import torch
    except ImportError:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001 - torch without a usable device: the engine reports the real error
        pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libpdengine.so and declare the prototypes of include/pdengine.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PDENGINE_LIB") or LIB_PATH    # PDENGINE_LIB: A/B another build of the library
    if not os.path.exists(p):
        raise PdError(f"{p} not found: build it with `make -C {_CSRC}` (or __graft_entry__.build()); "
                      "pdengine has no CPU fallback")
    _torch_runtime_first()
    lib = C.CDLL(p)
    lib.pd_last_error.restype = C.c_char_p
    lib.pd_abi_version.restype = C.c_int
    lib.pd_engine_create.argtypes = [C.POINTER(pd_config), C.c_int, C.POINTER(C.c_void_p)]
    lib.pd_engine_destroy.argtypes = [C.c_void_p]
    lib.pd_engine_destroy.restype = None
    lib.pd_param_count.argtypes = [C.c_void_p]
    lib.pd_param_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.pd_load_weights.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32]
    lib.pd_init_random_weights.argtypes = [C.c_void_p, C.c_uint64]
    lib.pd_weights_missing.argtypes = [C.c_void_p]
    lib.pd_vae_weights_missing.argtypes = [C.c_void_p]
    lib.pd_vae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.pd_text_weights_missing.argtypes = [C.c_void_p]
    lib.pd_text_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.pd_text_encode_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.pd_eps.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]
    lib.pd_control_shape.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32] + [C.POINTER(C.c_int32)] * 3
    lib.pd_ddim_sample.argtypes = [C.c_void_p, C.POINTER(pd_sample_args), C.c_int32, C.c_void_p, C.c_void_p]
    lib.pd_sample_begin.argtypes = [C.c_void_p, C.POINTER(pd_sample_args)]
    lib.pd_sample_step.argtypes = [C.c_void_p, C.c_int32]
    lib.pd_sample_get.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.pd_sample_set_latents.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.pd_sample_eps_at.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.pd_sample_set_guidance.argtypes = [C.c_void_p, C.c_float]
    lib.pd_sample_end.argtypes = [C.c_void_p]
    lib.pd_make_schedule.argtypes = [C.c_void_p, C.c_int32, C.c_float] + [C.c_void_p] * 5
    lib.pd_synchronize.argtypes = [C.c_void_p]
    lib.pd_stream.argtypes = [C.c_void_p]
    lib.pd_stream.restype = C.c_void_p
    lib.pd_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.pd_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.pd_comm_new_id.argtypes = [C.c_void_p]
    lib.pd_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.pd_comm_world.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pd_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.pd_comm_destroy.argtypes = [C.c_void_p]
    lib.pd_get_stat.argtypes = [C.c_void_p, C.c_char_p]
    lib.pd_get_stat.restype = C.c_int64
    lib.pd_bench_conv3x3.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.POINTER(C.c_float)]
    lib.pd_bench_linear.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float)]
    lib.pd_profile_dump.argtypes = [C.c_void_p, C.c_char_p]
    lib.pd_profile_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    # per-op parity hooks (include/pdengine_ops.h)
    fp = C.c_void_p
    lib.pd_op_conv2d.argtypes = [C.c_void_p, fp, fp, fp, fp] + [C.c_int] * 9 + [C.c_float, C.c_int, fp]
    lib.pd_op_linear.argtypes = [C.c_void_p, fp, fp, fp] + [C.c_int] * 5 + [fp]
    lib.pd_op_linear_fp8.argtypes = [C.c_void_p, fp, fp, fp] + [C.c_int] * 4 + [fp]
    lib.pd_op_groupnorm.argtypes = [C.c_void_p, fp, fp, fp] + [C.c_int] * 4 + [C.c_float, C.c_int, fp]
    lib.pd_op_layernorm.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.c_int, fp]
    lib.pd_op_attention.argtypes = [C.c_void_p, fp, fp, fp] + [C.c_int] * 4 + [fp]
    lib.pd_op_spatial_transformer.argtypes = [C.c_void_p, C.c_char_p, fp, fp] + [C.c_int] * 3 + [fp]
    lib.pd_op_time_embed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, fp, fp]
    if path is None:
        _lib = lib
    return lib


EXPORTS = [
    "pd_last_error", "pd_abi_version", "pd_engine_create", "pd_engine_destroy", "pd_param_count", "pd_param_info",
    "pd_load_weights", "pd_init_random_weights", "pd_weights_missing", "pd_vae_weights_missing", "pd_vae_decode", "pd_eps", "pd_control_shape", "pd_ddim_sample",
    "pd_sample_begin", "pd_sample_step", "pd_sample_get", "pd_sample_set_latents", "pd_sample_set_guidance", "pd_sample_eps_at", "pd_sample_end",
    "pd_make_schedule", "pd_synchronize", "pd_stream", "pd_wait_stream", "pd_set_option", "pd_get_stat", "pd_bench_conv3x3", "pd_bench_linear", "pd_text_encode", "pd_text_encode_ex", "pd_text_weights_missing",
    "pd_profile_read", "pd_profile_dump", "pd_comm_new_id", "pd_comm_init", "pd_comm_world", "pd_comm_all_gather", "pd_comm_destroy",
    "pd_sd3_configure", "pd_sd3_weights_missing", "pd_sd3_forward", "pd_sd3_control", "pd_sd3_sample", "pd_sd3_down_proj",
    "pd_op_conv2d", "pd_op_linear", "pd_op_linear_fp8", "pd_op_groupnorm", "pd_op_layernorm", "pd_op_attention", "pd_op_spatial_transformer", "pd_op_time_embed",
]


def make_config(cfg: ModelConfig, precision: int = PD_PREC_F16, stream_f32: bool = False) -> pd_config:
    c = pd_config()
    c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
    c.hint_channels, c.query_channels = cfg.hint_channels, cfg.query_channels
    c.model_channels = cfg.model_channels
    c.num_levels = len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        c.channel_mult[i] = m
    c.num_res_blocks = cfg.num_res_blocks
    c.num_attn_res = len(cfg.attention_resolutions)
    for i, m in enumerate(cfg.attention_resolutions):
        c.attention_resolutions[i] = m
    c.num_heads = cfg.num_heads
    c.context_dim, c.context_len = cfg.context_dim, cfg.context_len
    for i, m in enumerate(cfg.hint_widths):
        c.hint_widths[i] = m
    c.timesteps = cfg.timesteps
    c.linear_start, c.linear_end = cfg.linear_start, cfg.linear_end
    c.precision = precision
    c.stream_f32 = 1 if stream_f32 else 0
    c.vae_ch = cfg.vae_ch
    c.vae_num_levels = len(cfg.vae_ch_mult)
    for i, m in enumerate(cfg.vae_ch_mult):
        c.vae_ch_mult[i] = m
    c.vae_num_res_blocks, c.vae_out_ch, c.scale_factor = cfg.vae_num_res_blocks, cfg.vae_out_ch, cfg.scale_factor
    c.text_vocab, c.text_layers, c.text_heads, c.text_ff = cfg.text_vocab, cfg.text_layers, cfg.text_heads, cfg.text_ff
    return c


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _to_host(x):
    return x.detach().cpu().numpy() if _is_torch(x) else np.asarray(x)


class _Buf:
    """A float32 (or int64) contiguous buffer as (pointer, mem-space), keeping its owner alive."""

    def __init__(self, x, dtype=np.float32):
        if x is None:
            self.owner, self.ptr, self.mem = None, None, None
            return
        if _is_torch(x):
            import torch
            td = torch.float32 if dtype == np.float32 else torch.int64
            t = x.detach().to(td).contiguous()
            self.owner, self.ptr = t, t.data_ptr()
            self.mem = PD_MEM_DEVICE if t.is_cuda else PD_MEM_HOST
        else:
            a = np.ascontiguousarray(x, dtype=dtype)
            self.owner, self.ptr, self.mem = a, a.ctypes.data, PD_MEM_HOST


class Engine:
    """One engine <-> one GPU <-> one HIP stream (SURVEY.md §8b threading row)."""

    def __init__(self, cfg: ModelConfig = SD15, device: int = 0, precision: str = "f16", stream_f32: bool = False,
                 lib_path: Optional[str] = None):
        """precision: "f16" (default: the reference's own GPU dtype, README.md:44-45), "bf16", "f16x2" (split fp16
        operands over fp32 storage) or "f32"."""
        _torch_runtime_first()
        self.lib = load_library(lib_path)
        self.cfg = cfg
        self.precision = PRECISIONS[precision]
        self.device = device
        self._h = C.c_void_p()
        c = make_config(cfg, self.precision, stream_f32)
        self._check(self.lib.pd_engine_create(C.byref(c), device, C.byref(self._h)))
        self._keep: List[_Buf] = []

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise PdError(self.lib.pd_last_error().decode(errors="replace"))

    def _order_after_torch(self, mem: int, device=None) -> None:
        """CUDA tensors were produced on torch's current stream; the engine reads them on its own non-blocking streams.
        Make those wait for everything enqueued on the producer stream so far (incl. the .to()/.contiguous() copies _Buf
        just launched)."""
        if mem != PD_MEM_DEVICE:
            return
        import torch
        st = torch.cuda.current_stream(device)
        self._check(self.lib.pd_wait_stream(self._h, C.c_void_p(st.cuda_stream)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.pd_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def param_names(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        name, nd, shp = C.c_char_p(), C.c_int32(), (C.c_int64 * 4)()
        for i in range(self.lib.pd_param_count(self._h)):
            self._check(self.lib.pd_param_info(self._h, i, C.byref(name), C.byref(nd), shp))
            out.append((name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))))
        return out

    def load_tensor(self, name: str, array) -> None:
        if _is_torch(array):
            array = array.detach().float().cpu().numpy()
        a = np.ascontiguousarray(array)
        if a.dtype == np.float16:
            dt = PD_DT_F16
        else:
            a = a.astype(np.float32, copy=False)
            dt = PD_DT_F32
        shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
        self._check(self.lib.pd_load_weights(self._h, name.encode(), a.ctypes.data, shp, a.ndim, dt))

    def load_state_dict(self, items: Iterable[Tuple[str, np.ndarray]], strict: bool = True) -> None:
        """Walk a reference checkpoint (``model.diffusion_model.*`` / ``control_model.*`` keys,
        cldm/model.py:12-21); other prefixes (first_stage_model, cond_stage_model) are skipped."""
        known = {n for n, _ in self.param_names()}
        it = items.items() if isinstance(items, dict) else items
        for name, arr in it:
            if name in known:
                self.load_tensor(name, arr)
        if strict and self.weights_missing():
            raise PdError(f"{self.weights_missing()} tensors missing after load_state_dict")

    def init_random_weights(self, seed: int = 1234) -> None:
        self._check(self.lib.pd_init_random_weights(self._h, seed))

    def weights_missing(self) -> int:
        return int(self.lib.pd_weights_missing(self._h))

    def vae_weights_missing(self) -> int:
        return int(self.lib.pd_vae_weights_missing(self._h))

    def vae_decode(self, latents):
        """decode_first_stage (ddpm.py:820-828): latents [B,4,h,w] -> images [B,3,8h,8w] in about [-1, 1]."""
        b = _Buf(latents)
        B, _, h, w = b.owner.shape
        if b.mem == PD_MEM_DEVICE:
            import torch
            out = torch.empty((B, self.cfg.vae_out_ch, 8 * h, 8 * w), dtype=torch.float32, device=b.owner.device)
            op = out.data_ptr()
        else:
            out = np.empty((B, self.cfg.vae_out_ch, 8 * h, 8 * w), np.float32)
            op = out.ctypes.data
        self._order_after_torch(b.mem)
        self._check(self.lib.pd_vae_decode(self._h, b.ptr, B, h, w, b.mem, op))
        return out

    def text_weights_missing(self) -> int:
        return int(self.lib.pd_text_weights_missing(self._h))

    def text_encode(self, input_ids, clip_skip: int = 0):
        """FrozenCLIPEmbedder.forward after tokenisation (ldm/modules/encoders/modules.py:118-128): token ids
        [B, context_len] -> last_hidden_state [B, context_len, context_dim] fp32 (NumPy in, NumPy out; CUDA int32 tensor in,
        CUDA tensor out).  clip_skip k: hidden_states[-(k+1)] through final_layer_norm (pipeline_prompt_diffusion.py:398-413)."""
        clip_skip = int(clip_skip or 0)
        if _is_torch(input_ids):
            import torch
            ids = input_ids.to(torch.int32).contiguous()
            if ids.is_cuda:
                out = torch.empty(tuple(ids.shape) + (self.cfg.context_dim,), dtype=torch.float32, device=ids.device)
                self._order_after_torch(PD_MEM_DEVICE)
                self._check(self.lib.pd_text_encode_ex(self._h, ids.data_ptr(), ids.shape[0], PD_MEM_DEVICE, clip_skip, out.data_ptr()))
                return out
            input_ids = ids.numpy()
        ids = np.ascontiguousarray(input_ids, np.int32)
        if ids.ndim != 2 or ids.shape[1] != self.cfg.context_len:
            raise ValueError(f"input_ids must be [B, {self.cfg.context_len}]")
        out = np.empty(ids.shape + (self.cfg.context_dim,), np.float32)
        self._check(self.lib.pd_text_encode_ex(self._h, ids.ctypes.data, ids.shape[0], PD_MEM_HOST, clip_skip, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ operator boundary
    def control_shapes(self, h: int, w: int) -> List[Tuple[int, int, int]]:
        out = []
        c, hh, ww = C.c_int32(), C.c_int32(), C.c_int32()
        n = len([0 for _ in range(PD_NUM_CONTROL)])
        from .weights import encoder_layout
        n = len(encoder_layout(self.cfg)) + 1
        for i in range(n):
            self._check(self.lib.pd_control_shape(self._h, i, h, w, C.byref(c), C.byref(hh), C.byref(ww)))
            out.append((c.value, hh.value, ww.value))
        return out

    def eps(self, x, t, ctx, pair, query, scales: Optional[Sequence[float]] = None, return_control: bool = False):
        """apply_model (cldm/cldm.py:369-382): eps [Bf,4,h,w] (and the 13 scaled control tensors)."""
        xb, cb, pb, qb = _Buf(x), _Buf(ctx), _Buf(pair), _Buf(query)
        Bf, _, h, w = xb.owner.shape
        mem = xb.mem
        if any(b.mem != mem for b in (cb, pb, qb)):
            raise PdError("all inputs must live in the same memory space")
        tb = _Buf(t, np.int64)
        if tb.mem != mem:
            raise PdError("t must live in the same memory space as x")
        sc = None if scales is None else np.ascontiguousarray(scales, dtype=np.float32)
        shapes = self.control_shapes(h, w)
        if mem == PD_MEM_DEVICE:
            import torch
            eps = torch.empty((Bf, self.cfg.out_channels, h, w), dtype=torch.float32, device=xb.owner.device)
            res = torch.empty(sum(Bf * c * a * b for c, a, b in shapes), dtype=torch.float32,
                              device=xb.owner.device) if return_control else None
            ep, rp = eps.data_ptr(), (res.data_ptr() if res is not None else None)
        else:
            eps = np.empty((Bf, self.cfg.out_channels, h, w), np.float32)
            res = np.empty(sum(Bf * c * a * b for c, a, b in shapes), np.float32) if return_control else None
            ep, rp = eps.ctypes.data, (res.ctypes.data if res is not None else None)
        self._order_after_torch(mem)
        self._check(self.lib.pd_eps(self._h, xb.ptr, tb.ptr, cb.ptr, pb.ptr, qb.ptr,
                                    None if sc is None else sc.ctypes.data, Bf, h, w, mem, ep, rp))
        if not return_control:
            return eps
        outs, off = [], 0
        for c, a, b in shapes:
            n = Bf * c * a * b
            outs.append(res[off:off + n].reshape(Bf, c, a, b))
            off += n
        return eps, outs

    # ------------------------------------------------------------------ sampling
    def num_ddim_steps(self, steps: int) -> int:
        """len(make_ddim_timesteps('uniform')) -- may exceed `steps` (util.py:47-49)."""
        T = self.cfg.timesteps
        return len(range(0, T, T // steps))

    def make_schedule(self, steps: int, eta: float = 0.0) -> Dict[str, np.ndarray]:
        n = self.num_ddim_steps(steps)
        ts = np.zeros(n, np.int64)
        arrs = [np.zeros(n, np.float32) for _ in range(4)]
        self._check(self.lib.pd_make_schedule(self._h, steps, eta, ts.ctypes.data, *[a.ctypes.data for a in arrs]))
        return dict(ddim_timesteps=ts, ddim_alphas=arrs[0], ddim_alphas_prev=arrs[1], ddim_sigmas=arrs[2],
                    ddim_sqrt_one_minus_alphas=arrs[3])

    def _args(self, *, x_T, ctx_cond, ctx_uncond, pair, query, steps, cfg_scale, eta=0.0, use_cfg=True,
              guess_mode=False, only_mid_control=False, temperature=1.0, control_scales=None,
              control_scales_step=None, noise=None, pair_uncond=None, query_uncond=None, timesteps=None):
        bufs = dict(x_T=_Buf(x_T), ctx_cond=_Buf(ctx_cond), ctx_uncond=_Buf(ctx_uncond), pair=_Buf(pair),
                    query=_Buf(query), pair_uncond=_Buf(pair_uncond), query_uncond=_Buf(query_uncond), noise=_Buf(noise))
        mems = {b.mem for b in bufs.values() if b.mem is not None}
        if len(mems) != 1:
            raise PdError("all inputs must live in the same memory space (all NumPy or all CUDA tensors)")
        a = pd_sample_args()
        B, _, h, w = bufs["x_T"].owner.shape
        a.batch, a.h, a.w, a.steps = B, h, w, steps
        a.eta, a.cfg_scale, a.use_cfg = eta, cfg_scale, 1 if use_cfg else 0
        a.guess_mode, a.only_mid_control, a.temperature = int(guess_mode), int(only_mid_control), temperature
        a.mem = mems.pop()
        for k, b in bufs.items():
            setattr(a, k, b.ptr)
        keep = list(bufs.values())
        n_steps = self.num_ddim_steps(steps)
        if timesteps is not None:       # custom grid, sampling order (descending); a.steps = its length
            ts = np.ascontiguousarray(_to_host(timesteps), dtype=np.int64).reshape(-1)
            a.steps = n_steps = len(ts)
            a.timesteps = ts.ctypes.data
            keep.append(ts)
        if control_scales is not None:
            cs = np.zeros(PD_NUM_CONTROL, np.float32)
            cs[:len(control_scales)] = control_scales
            a.control_scales = cs.ctypes.data
            keep.append(cs)
        if control_scales_step is not None:
            css = np.ascontiguousarray(control_scales_step, dtype=np.float32)
            assert css.shape == (n_steps, PD_NUM_CONTROL)
            a.control_scales_step = css.ctypes.data
            keep.append(css)
        self._order_after_torch(a.mem)
        return a, keep, (B, h, w)

    def ddim_sample(self, *, return_intermediates: bool = False, **kw):
        """The fused loop (DDIMSampler.sample, cldm/ddim_hacked.py:55-178): returns latents [B,4,h,w]
        (NumPy, or a CUDA tensor when the inputs were CUDA tensors) and optionally x_inter [S+1,B,4,h,w]."""
        a, keep, (B, h, w) = self._args(**kw)
        S = a.steps if a.timesteps else self.num_ddim_steps(a.steps)
        Cc = self.cfg.in_channels
        if a.mem == PD_MEM_DEVICE:
            import torch
            dev = keep[0].owner.device
            out = torch.empty((B, Cc, h, w), dtype=torch.float32, device=dev)
            inter = torch.empty((S + 1, B, Cc, h, w), dtype=torch.float32, device=dev) if return_intermediates else None
            op, ip = out.data_ptr(), (inter.data_ptr() if inter is not None else None)
        else:
            out = np.empty((B, Cc, h, w), np.float32)
            inter = np.empty((S + 1, B, Cc, h, w), np.float32) if return_intermediates else None
            op, ip = out.ctypes.data, (inter.ctypes.data if inter is not None else None)
        self._check(self.lib.pd_ddim_sample(self._h, C.byref(a), a.mem, op, ip))
        del keep
        return (out, inter) if return_intermediates else out

    def sample_begin(self, **kw) -> int:
        a, keep, shape = self._args(**kw)
        self._check(self.lib.pd_sample_begin(self._h, C.byref(a)))
        self._keep = keep   # the staged copies stay alive until sample_end()
        self._ses = (shape, a.mem, keep[0].owner if a.mem == PD_MEM_DEVICE else None)
        return a.steps if a.timesteps else self.num_ddim_steps(a.steps)

    def sample_step(self, i: int) -> None:
        self._check(self.lib.pd_sample_step(self._h, i))

    def sample_get(self, what: int = PD_GET_LATENTS):
        (B, h, w), mem, like = self._ses
        if mem == PD_MEM_DEVICE:
            import torch
            out = torch.empty((B, self.cfg.in_channels, h, w), dtype=torch.float32, device=like.device)
            self._check(self.lib.pd_sample_get(self._h, what, mem, out.data_ptr()))
        else:
            out = np.empty((B, self.cfg.in_channels, h, w), np.float32)
            self._check(self.lib.pd_sample_get(self._h, what, mem, out.ctypes.data))
        return out

    def sample_set_latents(self, latents) -> None:
        b = _Buf(latents)
        self._order_after_torch(b.mem)
        self._check(self.lib.pd_sample_set_latents(self._h, b.mem, b.ptr))

    def sample_set_guidance(self, scale: float) -> None:
        """unconditional_guidance_scale of the following steps (ucg_schedule, cldm/ddim_hacked.py:159-161)."""
        self._check(self.lib.pd_sample_set_guidance(self._h, float(scale)))

    def sample_eps_at(self, t: int, scales: Optional[Sequence[float]] = None):
        sc = None
        if scales is not None:
            sc = np.zeros(PD_NUM_CONTROL, np.float32)
            sc[:len(scales)] = scales
        self._check(self.lib.pd_sample_eps_at(self._h, int(t), None if sc is None else sc.ctypes.data))
        return self.sample_get(PD_GET_EPS)

    def sample_end(self) -> None:
        self._check(self.lib.pd_sample_end(self._h))
        self._keep = []

    # ------------------------------------------------------------------ multi-GPU
    def comm_new_id(self) -> bytes:
        """The 128-byte rendezvous token rank 0 creates and hands to every rank (any host channel)."""
        buf = (C.c_uint8 * PD_COMM_ID_BYTES)()
        self._check(self.lib.pd_comm_new_id(buf))
        return bytes(buf)

    def comm_init(self, comm_id: bytes, world: int, rank: int) -> None:
        """Join the engine-owned RCCL communicator (collective over all `world` ranks; one process and one engine per GPU)."""
        if len(comm_id) != PD_COMM_ID_BYTES:
            raise ValueError("comm_id must be the %d bytes of comm_new_id()" % PD_COMM_ID_BYTES)
        buf = (C.c_uint8 * PD_COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._check(self.lib.pd_comm_init(self._h, buf, int(world), int(rank)))

    def comm_world(self) -> Tuple[int, int]:
        w, r = C.c_int32(), C.c_int32()
        self._check(self.lib.pd_comm_world(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_all_gather(self, latents):
        """[b, ...] from every rank -> [world * b, ...] in rank order (equal b on every rank); the path's only exchange."""
        world, _ = self.comm_world()
        b = _Buf(latents)
        self._order_after_torch(b.mem)
        shape = (world * b.owner.shape[0],) + tuple(b.owner.shape[1:])
        count = int(np.prod(b.owner.shape))
        if b.mem == PD_MEM_DEVICE:
            import torch
            out = torch.empty(shape, dtype=torch.float32, device=b.owner.device)
            self._check(self.lib.pd_comm_all_gather(self._h, b.ptr, out.data_ptr(), count, b.mem))
        else:
            out = np.empty(shape, np.float32)
            self._check(self.lib.pd_comm_all_gather(self._h, b.ptr, out.ctypes.data, count, b.mem))
        return out

    def comm_destroy(self) -> None:
        self._check(self.lib.pd_comm_destroy(self._h))

    # ------------------------------------------------------------------ instrumentation
    def synchronize(self):
        self._check(self.lib.pd_synchronize(self._h))

    def stat(self, key: str) -> int:
        return int(self.lib.pd_get_stat(self._h, key.encode()))

    def set_option(self, key: str, value: int):
        self._check(self.lib.pd_set_option(self._h, key.encode(), int(value)))

    def profile_read(self, klass: int = -1) -> Tuple[float, int, float]:
        """(device ms, launches, algorithmic FLOPs) of the launches bracketed while option 'profile' was on."""
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.pd_profile_read(self._h, klass, C.byref(ms), C.byref(n), C.byref(fl)))
        return float(ms.value), int(n.value), float(fl.value)

    def profile_dump(self, path: str) -> None:
        self._check(self.lib.pd_profile_dump(self._h, path.encode()))

    def bench_conv3x3(self, Bf: int, H: int, W: int, Cin: int, Cout: int, iters: int = 20) -> float:
        ms = C.c_float()
        self._check(self.lib.pd_bench_conv3x3(self._h, Bf, H, W, Cin, Cout, iters, C.byref(ms)))
        return float(ms.value)

    def bench_linear(self, M: int, K: int, N: int, residual: bool = False, iters: int = 20) -> float:
        ms = C.c_float()
        self._check(self.lib.pd_bench_linear(self._h, M, K, N, 1 if residual else 0, iters, C.byref(ms)))
        return float(ms.value)

    # ------------------------------------------------------------------ per-op parity hooks
    def op_conv2d(self, x, w, b=None, residual=None, stride=1, upsample=False, silu=False, scale=1.0, stream_out=False):
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
        B, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        Hv, Wv = (H * 2, W * 2) if upsample else (H, W)
        Ho, Wo = ((Hv + 1) // 2, (Wv + 1) // 2) if stride == 2 else (Hv, Wv)
        y = np.empty((B, Cout, Ho, Wo), np.float32)
        bb = None if b is None else np.ascontiguousarray(b, np.float32)
        rr = None if residual is None else np.ascontiguousarray(residual, np.float32)
        self._check(self.lib.pd_op_conv2d(self._h, x.ctypes.data, w.ctypes.data, None if bb is None else bb.ctypes.data,
                                          None if rr is None else rr.ctypes.data, B, Cin, H, W, Cout, k, stride,
                                          int(upsample), int(silu), scale, int(stream_out), y.ctypes.data))
        return y

    def op_linear(self, x, w, b=None, geglu=False, a_silu=False):
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
        M, K = x.shape
        N = w.shape[0] // 2 if geglu else w.shape[0]
        y = np.empty((M, N), np.float32)
        bb = None if b is None else np.ascontiguousarray(b, np.float32)
        self._check(self.lib.pd_op_linear(self._h, x.ctypes.data, w.ctypes.data, None if bb is None else bb.ctypes.data,
                                          M, K, N, int(geglu), int(a_silu), y.ctypes.data))
        return y

    def op_linear_fp8(self, x, w, b=None, gelu_tanh=False):
        """The SD3 path's e4m3 linear layer (per-row scales of x and w) on host arrays."""
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
        M, K = x.shape
        N = w.shape[0]
        y = np.empty((M, N), np.float32)
        bb = None if b is None else np.ascontiguousarray(b, np.float32)
        self._check(self.lib.pd_op_linear_fp8(self._h, x.ctypes.data, w.ctypes.data, None if bb is None else bb.ctypes.data, M, K, N,
                                              4 if gelu_tanh else 0, y.ctypes.data))
        return y

    def op_groupnorm(self, x, gamma, beta, eps=1e-5, silu=False):
        x = np.ascontiguousarray(x, np.float32)
        g = np.ascontiguousarray(gamma, np.float32); b = np.ascontiguousarray(beta, np.float32)
        B, Cc, H, W = x.shape
        y = np.empty_like(x)
        self._check(self.lib.pd_op_groupnorm(self._h, x.ctypes.data, g.ctypes.data, b.ctypes.data, B, Cc, H, W, eps,
                                             int(silu), y.ctypes.data))
        return y

    def op_layernorm(self, x, gamma, beta):
        x = np.ascontiguousarray(x, np.float32)
        g = np.ascontiguousarray(gamma, np.float32); b = np.ascontiguousarray(beta, np.float32)
        rows, Cc = x.shape
        y = np.empty_like(x)
        self._check(self.lib.pd_op_layernorm(self._h, x.ctypes.data, g.ctypes.data, b.ctypes.data, rows, Cc, y.ctypes.data))
        return y

    def op_time_embed(self, t, net: int = 0, want_emb: bool = True):
        """(timestep_embedding(t, model_channels), time_embed(...)) of the loaded UNet (net 0) / ControlNet (net 1)."""
        t = np.ascontiguousarray(t, np.int64)
        mc = self.cfg.model_channels
        temb = np.empty((t.size, mc), np.float32)
        emb = np.empty((t.size, 4 * mc), np.float32) if want_emb else None
        self._check(self.lib.pd_op_time_embed(self._h, int(net), t.ctypes.data, int(t.size), temb.ctypes.data, emb.ctypes.data if want_emb else None))
        return temb, emb

    def op_attention(self, q, k, v):
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        B, Nq, Cc = q.shape
        Nk = k.shape[1]
        o = np.empty_like(q)
        self._check(self.lib.pd_op_attention(self._h, q.ctypes.data, k.ctypes.data, v.ctypes.data, B, Nq, Nk, Cc, o.ctypes.data))
        return o

    def op_spatial_transformer(self, prefix: str, x, context):
        """SpatialTransformer.forward (attention.py:321-340) of the block loaded under `prefix`, through the sampling code path."""
        x = np.ascontiguousarray(x, np.float32); ctx = np.ascontiguousarray(context, np.float32)
        B, Cc, H, W = x.shape
        y = np.empty_like(x)
        self._check(self.lib.pd_op_spatial_transformer(self._h, prefix.encode(), x.ctypes.data, ctx.ctypes.data, B, H, W, y.ctypes.data))
        return y
