"""ctypes binding of libstdiff_hip.so / libstdiff_hip_f16.so (include/stdiff_hip.h).  Fails loudly if a library
is missing: there is no CPU or eager-PyTorch fallback anywhere in this package.

The two libraries are the same sources built with bf16 resp. IEEE fp16 as the 16-bit MFMA operand format; at the
C ABI both call it dtype code 1.  Python keeps them apart by the torch dtype of the tensors (``lib_for``)."""
import ctypes as C
import os

import torch

F32, BF16, F16 = 0, 1, 2         # F16 exists on the Python side only: it selects the fp16 build, whose C code is 1
EPI_BIAS, EPI_SILU, EPI_GELU, EPI_GATE_RES, EPI_ADD, EPI_GELU_SAVE, EPI_ACCUM, EPI_LN_MOD, EPI_GELU_BWD = range(9)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstdiff_hip.so")
LIB_PATH_F16 = os.path.join(_HERE, "libstdiff_hip_f16.so")

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float


class SplitResult(C.Structure):
    """stdiff_split_result: a GEMM result left as split-K partial products (slices == 0: the GEMM completed)."""
    _fields_ = [("ws", vp), ("ldws", i32), ("slices", i32), ("slice_stride", C.c_longlong)]


class GemmArgs(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32),
                ("A", vp), ("lda", i32), ("a_dtype", i32),
                ("W", vp), ("ldw", i32),
                ("bias", vp),
                ("C", vp), ("ldc", i32), ("c_dtype", i32),
                ("epi", i32),
                ("gate", vp), ("ldgate", i32),
                ("res", vp), ("ldres", i32),
                ("C2", vp), ("ldc2", i32),
                ("workspace", vp), ("workspace_bytes", C.c_longlong),
                ("a_layout", i32), ("w_layout", i32), ("k_valid", i32),
                ("ln_parts", vp), ("ln_nparts", i32), ("stats_out", vp), ("defer", C.POINTER(SplitResult))]


class StepArgs(C.Structure):
    _fields_ = [("x", vp), ("eps", vp), ("gt", vp), ("mask", vp), ("coef", vp),
                ("t_dev", vp), ("t_first", i32),
                ("z_base", vp), ("seed", u64), ("row_offset", i64),
                ("xp", vp), ("ldxp", i32), ("xp_dtype", i32),
                ("R", i32), ("G", i32), ("pred_type", i32), ("z_rows", i64)]


class BlockWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                  "ada_w", "ada_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "gate_w", "gate_b")]


class DitWeights(C.Structure):
    _fields_ = [("dtype", i32), ("G", i32), ("Gp", i32), ("H", i32), ("heads", i32), ("depth", i32),
                ("in_w", vp), ("in_b", vp), ("out_w", vp), ("out_b", vp), ("fada_w", vp), ("fada_b", vp),
                ("blocks", C.POINTER(BlockWeights)), ("lnf_w", vp), ("lnf_b", vp)]


class DitWorkspace(C.Structure):
    _fields_ = [("R", i32), ("n_chunks", i32), ("max_chunk_rows", i32), ("cu_rows", vp),
                ("h", vp), ("mod", vp), ("xn", vp), ("qkv", vp), ("o", vp), ("u", vp), ("fmod", vp),
                ("ln_parts", vp), ("gates", vp)]


class TrainLinear(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("gw", vp), ("gb", vp)]


class TrainBlock(C.Structure):
    _fields_ = [(n, TrainLinear) for n in ("qkv", "proj", "fc1", "fc2", "ada")]


class TrainWeights(C.Structure):
    _fields_ = [("dtype", i32), ("G", i32), ("Gp", i32), ("H", i32), ("heads", i32), ("depth", i32)] + \
               [(n, TrainLinear) for n in ("t0", "t2", "cond", "in_", "fada", "out")] + \
               [("blocks", C.POINTER(TrainBlock)), ("freqs", vp)]


class TrainBlockActs(C.Structure):
    _fields_ = [(n, vp) for n in ("h", "hmid", "mod", "xn1", "xn2", "qkv", "o", "lse", "pa", "pm", "upre", "u",
                                  "mean1", "rstd1", "mean2", "rstd2")]


class GradRanges(C.Structure):
    """struct stdiff_grad_ranges: the gradient-norm pass folded into the backward, one range per gradient bucket."""
    _fields_ = [("sumsq", vp), ("lo", vp * 64), ("count", C.c_longlong * 64), ("in_group", i32), ("parts", vp),
                ("parts_count", C.c_longlong)]


class TrainActs(C.Structure):
    _fields_ = [("C", i32), ("cu_rows", vp)] + \
               [(n, vp) for n in ("xp", "yp", "e0", "a1", "e1", "temb", "c", "s")] + \
               [("blocks", C.POINTER(TrainBlockActs))] + \
               [(n, vp) for n in ("h_last", "fmod", "xnf", "meanf", "rstdf", "eps", "deps", "dh", "dxn")] + \
               [("dmod", vp * 2)] + \
               [(n, vp) for n in ("dfmod", "dbr", "du", "dupre", "do_", "dqkv", "delta", "ds32", "dc", "dcT", "de1",
                                  "da1", "dhT")] + \
               [("tr", (vp * 2) * 2), ("splitk_ws", vp), ("splitk_bytes", C.c_longlong),
                ("splitk_ws_side", vp), ("splitk_side_bytes", C.c_longlong), ("dbr_b", vp * 3), ("dqkv_b", vp),
                ("no_defer", i32), ("param_ready", C.POINTER(vp)),
                ("grad_ranges", C.POINTER(GradRanges))]


class DwProblem(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("dY", vp), ("lddy", i32), ("X", vp), ("ldx", i32), ("gw", vp), ("ldgw", i32),
                ("gb", vp)]


class DwGroupArgs(C.Structure):
    _fields_ = [("n", i32), ("C", i32), ("dtype", i32), ("p", DwProblem * 8), ("sq_parts", vp)]


BucketFn = C.CFUNCTYPE(None, i32, vp)


class AdamWArgs(C.Structure):
    _fields_ = [("params", vp), ("grads", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("n", i64),
                ("sumsq", vp), ("step_dev", vp), ("scal", vp),
                ("max_norm", f32), ("grad_scale", f32), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32),
                ("weight_decay", f32), ("bf16_mirror", vp)]


_SIGS = {
    "stdiff_abi_version": (i32, []),
    "stdiff_h16_format": (i32, []),
    "stdiff_error_string": (C.c_char_p, [i32]),
    "stdiff_reload_env": (None, []),
    "stdiff_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "stdiff_gemm_static_scratch": (i32, [C.POINTER(GemmArgs), vp]),
    "stdiff_dw_group": (i32, [C.POINTER(DwGroupArgs), vp]),
    "stdiff_dw_group_parts": (C.c_longlong, [C.POINTER(DwGroupArgs)]),
    "stdiff_sum_parts": (i32, [vp, C.c_longlong, vp, i32, vp]),
    "stdiff_transpose": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "stdiff_attention_fwd": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "stdiff_ln_modulate": (i32, [vp, i32, vp, vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, vp]),
    "stdiff_row_stats": (i32, [vp, i32, vp, i32, i32, vp]),
    "stdiff_silu_cond": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "stdiff_timestep_embedding": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "stdiff_pack_rows": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "stdiff_gather_rows2": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, vp]),
    "stdiff_column_moments": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "stdiff_column_js": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, vp]),
    "stdiff_rank_corr": (i32, [vp, vp, i64, vp, i32, i32, i32, vp]),
    "stdiff_preprocess_rows": (i32, [vp, i32, vp, i32, i32, i32, f32, f32, vp]),
    "stdiff_ddpm_step": (i32, [C.POINTER(StepArgs), vp]),
    "stdiff_advance_timestep": (i32, [vp, vp]),
    "stdiff_spin_probe": (i32, [i32, i32, vp]),
    "stdiff_init_sample": (i32, [vp, vp, u64, i64, vp, vp, vp, i32, i32, i32, i32, vp]),
    "stdiff_add_noise_splice": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "stdiff_masked_mse": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp]),
    "stdiff_ln_modulate_split": (i32, [C.POINTER(SplitResult), vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, vp,
                                       i32, i32, vp, vp, i32, i32, vp]),
    "stdiff_ln_modulate_bwd_split": (i32, [C.POINTER(SplitResult), i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32,
                                           vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "stdiff_attention_bwd_split": (i32, [vp, i32, vp, i32, C.POINTER(SplitResult), vp, i32, vp, vp, vp, i32, vp, i32, i32,
                                         i32, i32, i32, i32, vp]),
    "stdiff_attention_bwd": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stdiff_ln_modulate_bwd": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32, i32, i32, vp]),
    "stdiff_ln_modulate_bwd_gate": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32,
                                          vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "stdiff_gate_bwd": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "stdiff_gelu_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "stdiff_silu_bwd": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "stdiff_colsum": (i32, [vp, i32, i32, vp, i32, i32, i32, vp]),
    "stdiff_grad_sumsq": (i32, [vp, i64, vp, vp]),
    "stdiff_clip_adamw": (i32, [C.POINTER(AdamWArgs), vp]),
    "stdiff_adamw_prepare": (i32, [C.POINTER(AdamWArgs), vp]),
    "stdiff_adamw_range": (i32, [C.POINTER(AdamWArgs), i64, i64, vp]),
    "stdiff_dit_forward": (i32, [C.POINTER(DitWeights), C.POINTER(DitWorkspace), vp, vp, vp, i32, vp]),
    "stdiff_dit_train_forward": (i32, [C.POINTER(TrainWeights), C.POINTER(TrainActs), vp, vp, vp]),
    "stdiff_dit_backward": (i32, [C.POINTER(TrainWeights), C.POINTER(TrainActs), vp, BucketFn, vp, vp]),
    "stdiff_dit_backward_parts": (C.c_longlong, [C.POINTER(TrainWeights)]),
    "stdiff_graph_begin": (i32, [vp]),
    "stdiff_graph_end": (i32, [vp, C.POINTER(vp)]),
    "stdiff_graph_launch": (i32, [vp, vp]),
    "stdiff_graph_destroy": (i32, [vp]),
}

_libs = {}


def lib(fp16=False):
    """The loaded library (raises if it was not built: run `python -c 'import __graft_entry__ as g; g.build()'`).
    fp16=True: the build whose 16-bit operand format is IEEE fp16."""
    L = _libs.get(bool(fp16))
    if L is None:
        path = LIB_PATH_F16 if fp16 else LIB_PATH
        path = os.environ.get("STDIFF_LIB_F16" if fp16 else "STDIFF_LIB", path)     # A/B timing of another build
        base = os.path.basename(path)
        if not os.path.isfile(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension was not built (make -C stdiff_amd/csrc). "
                "stdiff_amd has no CPU fallback.")
        L = C.CDLL(path)
        missing = [name for name in _SIGS if not hasattr(L, name)]
        if missing:
            raise RuntimeError(f"{base} lacks symbols declared in include/stdiff_hip.h: {missing}")
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.stdiff_abi_version() != 14:
            raise RuntimeError(f"{base} ABI version mismatch")
        if L.stdiff_h16_format() != int(bool(fp16)):
            raise RuntimeError(f"{base} was built with the wrong 16-bit operand format")
        _libs[bool(fp16)] = L
    return L


def reload_env():
    """Make the loaded libraries re-read their STDIFF_GEMM_* / STDIFF_SPLITK_* knobs (they read the environment once)."""
    for L in _libs.values():
        L.stdiff_reload_env()


def lib_for(*tensors):
    """The library that computes in the 16-bit format of these tensors: the fp16 build if any of them is
    torch.float16 (mixing it with bfloat16 is an error), else the default (bf16 / fp32) build."""
    kinds = {t.dtype for t in tensors if isinstance(t, torch.Tensor)}
    if torch.float16 in kinds:
        if torch.bfloat16 in kinds:
            raise TypeError("float16 and bfloat16 tensors in one stdiff_hip call")
        return lib(True)
    return lib()


class StdiffError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().stdiff_error_string(rc).decode()
        if rc in (-1, -2, -3, -5):
            raise ValueError(f"stdiff_hip: {msg} (status {rc})")
        raise StdiffError(f"stdiff_hip: {msg} (status {rc})")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype in (torch.bfloat16, torch.float16):     # the C ABI's "16-bit operand" code; lib_for() picks the build
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(dt):
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[dt]


def c_dtype(dt):
    """Python compute-dtype code -> the C ABI's dtype code."""
    return F32 if dt == F32 else BF16


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("stdiff_amd runs on MI355X only: got a CPU tensor (no CPU fallback exists)")
