"""ctypes binding of libvcad_hip.so (include/vcad.h).  There is NO fallback: if the HIP library is missing
or fails to load, importing the native path raises — the product never computes on the CPU.

Build:  make -C videocad_amd/csrc        (or `python __graft_entry__.py`, which calls build())
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvcad_hip.so")
# the same sources built with fp16 as the 16-bit storage format (include/vcad.h: vcad_storage_format) — VCAD_F16 engines
LIB_PATH_F16 = os.path.join(_HERE, "csrc", "libvcad_hip_f16.so")

VCAD_F32, VCAD_BF16, VCAD_BF16X3, VCAD_F16 = 0, 1, 2, 3
# kernel-selection flags (include/vcad.h VCAD_GEMM_*; tests only) and kernel families (vcad_kernel_launches / vcad_op_gemm kernel_out)
GEMM_TILE64, GEMM_TILE128, GEMM_DMA_NEVER, GEMM_DMA_ALWAYS, GEMM_WIDE_NEVER, GEMM_WIDE_ALWAYS, GEMM_MID_NEVER, GEMM_MID_ALWAYS = 1, 2, 4, 8, 16, 32, 64, 128
GEMM_DYNAMIC = 1 << 16
GEMM_MINI_NEVER, GEMM_MINI_ALWAYS = 1 << 21, 1 << 22      # persistent kernel: mini tiles for the rows of a mostly empty last round (automatic otherwise)


def gemm_reserve_cus(n8):
    """persistent GEMM on 256 - 8 * n8 CUs (n8 = 0..15): the rest stay free for other streams' kernels (RCCL during the gradient exchange)"""
    return (int(n8) & 15) << 17

KERNEL_GEMM_DMA, KERNEL_GEMM_REG, KERNEL_GEMM_MID, KERNEL_GEMM_GROUPED = 1, 2, 3, 4


def gemm_xcd_cols(n):
    return int(n) << 8


def gemm_ngroup(n):
    return int(n) << 12

NMETRIC = 32

# metric slots (csrc/loss.h)
MET_CMD_CORRECT, MET_CMD_COUNT, MET_PAR_CORRECT, MET_PAR_COUNT = 0, 5, 10, 16
MET_CMD_CORRECT_TOPK, MET_CMD_COUNT_TOPK, MET_PAR_CORRECT_TOPK, MET_PAR_COUNT_TOPK, MET_CORRECT, MET_TOTAL = 22, 23, 24, 25, 26, 27


class Config(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "hidden_size", "nhead", "num_decoder_layers", "dim_feedforward", "window_size", "act_dim",
        "num_classes", "num_params", "num_params_values", "max_ep_len",
        "vit_dim", "vit_depth", "vit_heads", "vit_dim_head", "vit_mlp", "image_size", "patch_size", "dtype",
        "enable_past_actions", "enable_past_states", "enable_timestep_embedding", "num_views")]


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

PROTOTYPES = {
    "vcad_last_error": (C.c_char_p, []),
    "vcad_version": (C.c_char_p, []),
    "vcad_storage_format": (C.c_char_p, []),
    "vcad_set_grad_scale": (_i, [_vp, _f]),
    "vcad_grad_scale": (_f, [_vp]),
    "vcad_set_defer_unscale": (_i, [_vp, _i]),
    "vcad_engine_create": (_i, [C.POINTER(Config), C.POINTER(_vp)]),
    "vcad_engine_destroy": (None, [_vp]),
    "vcad_param_total": (_i64, [_vp]),
    "vcad_param_count": (_i, [_vp]),
    "vcad_param_info": (_i, [_vp, _i, C.c_char_p, _sz, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64 * 4), C.POINTER(_i)]),
    "vcad_bucket_count": (_i, [_vp]),
    "vcad_bucket_range": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "vcad_bind": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "vcad_sync_shadow": (_i, [_vp, _vp]),
    "vcad_workspace_bytes": (_sz, [_vp, _i, _i]),
    "vcad_set_workspace": (_i, [_vp, _vp, _sz]),
    "vcad_set_dropout": (_i, [_vp, _f, C.c_uint64]),
    "vcad_set_fp8": (_i, [_vp, _i]),
    "vcad_set_multiview": (_i, [_vp, _vp]),
    "vcad_dropout_mask": (_i, [_vp, _i, _i, _i, _i64, _vp]),
    "vcad_dropout_mask_range": (_i, [_vp, _i, _i, _i, _i64, _i64, _vp]),
    "vcad_forward": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "vcad_forward_u8": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "vcad_forward_rgb8": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "vcad_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float * 5), _vp, _vp, _vp, _vp]),
    "vcad_dlogits_offsets": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "vcad_backward": (_i, [_vp, _vp, _vp, _vp]),
    "vcad_backward_stage": (_i, [_vp, _i, _vp, _vp, _vp]),
    "vcad_backward_stage_side": (_i, [_vp, _i, _vp, _vp, _vp]),
    "vcad_side_stage": (_i, [_vp]),
    "vcad_set_bucket_callback": (_i, [_vp, _vp, _vp]),
    "vcad_join_side": (_i, [_vp, _vp]),
    "vcad_wire_amax": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "vcad_wire_pack": (_i, [_vp, _i64, _i64, _vp, _vp, _i, _vp]),
    "vcad_wire_unpack": (_i, [_vp, _i64, _i64, _vp, _vp, _i, _vp]),
    "vcad_optimizer_step": (_i, [_vp, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "vcad_optimizer_step_groups": (_i, [_vp, C.POINTER(_f), _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "vcad_infer_workspace_bytes": (_sz, [_vp, _i, _i]),
    "vcad_infer_begin": (_i, [_vp, _vp, _i, _i, _vp]),
    "vcad_infer_begin_u8": (_i, [_vp, _vp, _i, _i, _vp]),
    "vcad_infer_step": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "vcad_profile_begin": (None, []),
    "vcad_profile_kernel": (_i, [_i, C.POINTER(C.c_double * 4)]),
    "vcad_profile_end": (_i, [C.POINTER(C.c_double * 8), C.POINTER(C.c_double * 8), C.POINTER(C.c_double * 8), C.POINTER(C.c_int * 8)]),
    "vcad_set_gemm_flags": (_i, [_vp, C.c_uint32]),
    "vcad_kernel_launches": (_i64, [_vp, _i]),
    "vcad_set_side_stream": (_i, [_vp, _i]),
    "vcad_op_gemm": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _vp, _i, _vp, _i64, _f, _vp, _sz, C.c_uint32, C.POINTER(_i), _vp]),
    "vcad_op_wgrad_batched": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), _i, _vp, _sz, C.c_uint32, _vp]),
    "vcad_op_pack_x3": (_i, [_vp, _vp, _i64, _vp]),
    "vcad_op_quant_mx8": (_i, [_i, _vp, _i64, _vp, _vp, _i64, _i, _vp]),
    "vcad_op_gemm_mx8": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp, _i, _vp, _i64, _vp]),
    "vcad_op_layernorm_fwd": (_i, [_i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _f, _vp]),
    "vcad_op_layernorm_bwd": (_i, [_i, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "vcad_op_attention_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "vcad_op_attention_bwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                   _i, _i, _i, _i, _i, _i, _f, _vp]),
    "vcad_op_cls_attention_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "vcad_op_cls_attention_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _f, _vp]),
    "vcad_op_attention_bwd_o": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                     _i, _i, _i, _i, _i, _i, _f, _vp]),
}


# A/B build only (csrc/ab.h; `make -C videocad_amd/csrc ab`): process-global selectors used by tools/, absent from libvcad_hip.so
AB_PROTOTYPES = {name: (None, [_i]) for name in (
    "vcad_debug_force_gemm_tile", "vcad_debug_gemm_dma", "vcad_debug_gemm_wide", "vcad_debug_gemm_mid", "vcad_debug_gemm_xcd_cols",
    "vcad_debug_attn_variant", "vcad_debug_gemm_waves", "vcad_debug_split_gelu", "vcad_debug_no_side_stream", "vcad_debug_gemm_policy",
    "vcad_debug_gemm_epilogue", "vcad_debug_gemm_variant", "vcad_debug_gemm_stagger", "vcad_debug_gemm_skip", "vcad_debug_res_in_ln", "vcad_debug_wgrad_bk32", "vcad_debug_cls_path", "vcad_debug_frame_first", "vcad_debug_pe_fold", "vcad_debug_dec_h16", "vcad_debug_attn_prefetch", "vcad_debug_splitk_r06", "vcad_debug_batch_wgrad")}
AB_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_bin", "libvcad_ab.so")


def declare(lib):
    """Attach argtypes/restype for every symbol of include/vcad.h (raises AttributeError if one is missing)."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None
_lib_f16 = None


def load(fmt: str = "bf16"):
    """Load the HIP library that stores `fmt` ("bf16": libvcad_hip.so, "f16": libvcad_hip_f16.so); fail loudly (no CPU fallback)."""
    global _lib, _lib_f16
    if fmt not in ("bf16", "f16"):
        raise ValueError(f"storage format {fmt!r}: expected 'bf16' or 'f16'")
    cur = _lib if fmt == "bf16" else _lib_f16
    if cur is None:
        import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch ships, not load a second one from /opt/rocm)
        path = LIB_PATH if fmt == "bf16" else LIB_PATH_F16
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `make -C videocad_amd/csrc` "
                               "(the VideoCAD MI355X path has no CPU fallback)")
        cur = declare(C.CDLL(path))
        got = cur.vcad_storage_format().decode()
        if got != fmt:
            raise RuntimeError(f"{path} stores {got!r}, expected {fmt!r}: rebuild with `make -C videocad_amd/csrc`")
        if fmt == "bf16":
            _lib = cur
        else:
            _lib_f16 = cur
    return cur


def storage_format(dtype: int) -> str:
    """which build of the library an engine of this dtype runs on"""
    return "f16" if dtype == VCAD_F16 else "bf16"


def load_ab():
    """The A/B build, for the measurement scripts under tools/ only: becomes the library the package runs on in THIS process (call it
    before anything else loads).  Never used by the product path or the tests."""
    global _lib
    if _lib is not None and getattr(_lib, "_vcad_ab", False):
        return _lib
    if _lib is not None:
        raise RuntimeError("load_ab() must run before the product library is loaded")
    import torch  # noqa: F401  (see load())
    path = os.environ.get("VCAD_AB_LIB", AB_LIB_PATH)          # (an older A/B build kept beside the current one, for before / after timings)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `make -C videocad_amd/csrc ab`")
    lib = C.CDLL(path)
    for name, (res, args) in list(PROTOTYPES.items()) + list(AB_PROTOTYPES.items()):
        if not hasattr(lib, name) and "VCAD_AB_LIB" in os.environ:
            continue                                           # an older build may lack newer entry points
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.vcad_debug_hog.restype = _i
    lib.vcad_debug_hog.argtypes = [_i, _i, _vp, _vp]
    lib._vcad_ab = True
    _lib = lib
    return lib


def check(lib, rc, what=""):
    if rc != 0:
        msg = lib.vcad_last_error()
        raise RuntimeError(f"libvcad {what} failed (code {rc}): {msg.decode() if msg else ''}")
