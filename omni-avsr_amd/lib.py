"""ctypes binding of libomniavsr_hip.so (C ABI: include/omni_avsr.h).  Loading fails loudly when the shared
library has not been built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C omni-avsr_amd/csrc`);
there is no fallback path of any kind."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OAVSR_LIB points the binding at another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("OAVSR_LIB") or os.path.join(_HERE, "libomniavsr_hip.so")

ABI_VERSION = 4      # OAVSR_ABI_VERSION of include/omni_avsr.h this binding's structures and argument lists follow
MAX_RATES = 8
GEMM_SYNC_INTS = 4096      # OAVSR_GEMM_SYNC_INTS: zeroed ints a split-K oavsr_op_gemm launch needs (GemmDesc.sync)
DT_F32, DT_BF16 = 0, 1
TASK_ID = {"audio": 0, "video": 1, "audiovisual": 2}


class OavsrConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("llm_family", C.c_int32),
        ("llm_hidden", C.c_int32), ("llm_layers", C.c_int32), ("llm_heads", C.c_int32), ("llm_kv_heads", C.c_int32),
        ("llm_head_dim", C.c_int32), ("llm_mlp", C.c_int32), ("llm_vocab", C.c_int32),
        ("llm_rms_eps", C.c_float), ("llm_qkv_bias", C.c_int32),
        ("lora_inner", C.c_int32), ("lora_k2", C.c_int32), ("lora_task_specific", C.c_int32),
        ("wh_dim", C.c_int32), ("wh_layers", C.c_int32), ("wh_heads", C.c_int32), ("wh_ffn", C.c_int32),
        ("av_dim", C.c_int32), ("av_layers", C.c_int32), ("av_heads", C.c_int32), ("av_ffn", C.c_int32),
        ("av_conv_pos", C.c_int32), ("av_conv_groups", C.c_int32), ("av_lora_k2", C.c_int32),
        ("proj_inter", C.c_int32), ("compression_stack", C.c_int32),
        ("n_rates_audio", C.c_int32), ("rates_audio", C.c_int32 * MAX_RATES),
        ("n_rates_video", C.c_int32), ("rates_video", C.c_int32 * MAX_RATES),
        ("proj_per_rate", C.c_int32), ("proj_layernorm_audio", C.c_int32), ("proj_layernorm_video", C.c_int32),
        ("id_bos", C.c_int32), ("id_eos", C.c_int32), ("id_pad", C.c_int32), ("id_audio_sos", C.c_int32),
        ("id_audio_eos", C.c_int32), ("id_video_sos", C.c_int32), ("id_video_eos", C.c_int32),
        ("prompt_len", C.c_int32 * 3),
    ]


class OavsrTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("nbytes", C.c_uint64)]


class OavsrProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("cu_ms", C.c_double)]


class OavsrTraceRec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("id", C.c_int32), ("steps", C.c_int32), ("rows", C.c_int32), ("t0_ms", C.c_float), ("t1_ms", C.c_float)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K2", C.c_int32),
        ("A", C.c_void_p), ("W", C.c_void_p), ("A2", C.c_void_p), ("W2", C.c_void_p),
        ("lda", C.c_int64), ("ldw", C.c_int64), ("lda2", C.c_int64), ("ldw2", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("out_f32", C.c_int32),
        ("col_scale", C.c_void_p), ("col_bias", C.c_void_p), ("prelu", C.c_void_p),
        ("act", C.c_int32), ("residual", C.c_void_p), ("ldr", C.c_int64), ("res_after_act", C.c_int32),
        ("groups", C.c_int32), ("a_goff", C.c_int32), ("amode", C.c_int32),
        ("H", C.c_int32), ("W_", C.c_int32), ("Cin", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("ksize", C.c_int32),
        ("rowinfo", C.c_void_p), ("taps", C.c_int32), ("Kvalid", C.c_int32), ("frameinfo", C.c_void_p),
        ("kernel", C.c_int32), ("swiglu_pairs", C.c_int32), ("decode_tiled", C.c_int32),
        ("ksplit", C.c_int32), ("partial", C.c_void_p), ("sync", C.c_void_p),
        ("row_sumsq", C.c_void_p), ("row_sumsq_tiles", C.c_int32), ("row_dim", C.c_int32), ("row_eps", C.c_float),
        ("ng_out", C.c_void_p), ("ng_ld", C.c_int64), ("ng_gain", C.c_void_p), ("ng_rowsq", C.c_void_p),
        ("grid", C.c_int32), ("rowsq_strips", C.c_int32), ("row_sumsq_form", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("head_dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("n_seqs", C.c_int32), ("max_q", C.c_int32), ("causal", C.c_int32),
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("cu_q", C.c_void_p), ("cu_k", C.c_void_p), ("scale", C.c_float), ("form", C.c_int32),
    ]


class DecodeAttnDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("head_dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("rows", C.c_int32), ("beams", C.c_int32), ("max_prefix", C.c_int32), ("max_tail", C.c_int32),
        ("Q", C.c_void_p), ("Kp", C.c_void_p), ("Vp", C.c_void_p), ("Kt", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
        ("ldq", C.c_int64), ("ldo", C.c_int64),
        ("prefix_len", C.c_void_p), ("step", C.c_void_p), ("anc", C.c_void_p), ("scale", C.c_float),
        ("inv_freq", C.c_void_p), ("pos", C.c_void_p), ("cpos", C.c_void_p),
    ]


# every symbol include/omni_avsr.h declares: (name, restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "oavsr_abi_version": (C.c_int, []),
    "oavsr_last_error": (C.c_char_p, []),
    "oavsr_env_report": (C.c_int, [C.c_char_p, _I]),
    "oavsr_set_launch_objective": (C.c_int, [C.c_int32]),
    "oavsr_get_launch_objective": (C.c_int, []),
    "oavsr_model_create": (C.c_int, [C.POINTER(OavsrConfig), _P, C.c_uint64, C.POINTER(OavsrTensor), _I, C.POINTER(_P)]),
    "oavsr_model_destroy": (None, [_P]),
    "oavsr_model_serialize_encoders": (C.c_int, [_P, C.c_int32]),
    "oavsr_workspace_create": (C.c_int, [_P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "oavsr_workspace_destroy": (None, [_P]),
    "oavsr_workspace_bytes": (C.c_uint64, [_P]),
    "oavsr_workspace_set_video_format": (C.c_int, [_P, _I, _I, _I]),
    "oavsr_workspace_set_digests": (C.c_int, [_P, _P]),
    "oavsr_encode_video": (C.c_int, [_P, _P, _P, C.POINTER(_I), _I, _I, _P, _P, _P]),
    "oavsr_encode_audio": (C.c_int, [_P, _P, _P, _L, C.POINTER(_I), C.POINTER(_I), _I, _I, _P, _P, _P, _P]),
    "oavsr_project": (C.c_int, [_P, _P, _I, _I, _P, _I, _P, _P]),
    "oavsr_build_prompt": (C.c_int, [_P, _P, _I, _I, C.POINTER(_I), _P, C.POINTER(_I), _P, _P, C.POINTER(_I), _P]),
    "oavsr_prefill": (C.c_int, [_P, _P, _I, _I, _P, C.POINTER(_I), _P, _P]),
    "oavsr_decode": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "oavsr_generate": (C.c_int, [_P, _P, _I, _P, C.POINTER(_I), _P, _L, C.POINTER(_I), C.POINTER(_I), _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "oavsr_generate_chunked": (C.c_int, [_P, _P, _I, _P, C.POINTER(_I), _P, _L, C.POINTER(_I), C.POINTER(_I), _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "oavsr_scheduler_create": (C.c_int, [_P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "oavsr_scheduler_destroy": (None, [_P]),
    "oavsr_scheduler_bytes": (C.c_uint64, [_P]),
    "oavsr_scheduler_caller_stream": (C.c_void_p, [_P, _I]),
    "oavsr_generate_submit": (C.c_int, [_P, _P, _P, _I, _P, C.POINTER(_I), _P, _L, C.POINTER(_I), C.POINTER(_I), _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, C.POINTER(_L)]),
    "oavsr_generate_wait": (C.c_int, [_P, _L, _P]),
    "oavsr_generate_poll": (C.c_int, [_P, _L]),
    "oavsr_scheduler_flush": (C.c_int, [_P]),
    "oavsr_scheduler_set_stride": (C.c_int, [_P, _I]),
    "oavsr_scheduler_trace": (C.c_int, [_P, _I]),
    "oavsr_scheduler_trace_read": (C.c_int, [_P, C.POINTER(OavsrTraceRec), _I]),
    "oavsr_scheduler_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "oavsr_profile_enable": (C.c_int, [_I]),
    "oavsr_profile_reset": (C.c_int, []),
    "oavsr_profile_read": (C.c_int, [C.POINTER(OavsrProfEntry), _I]),
    "oavsr_op_gemm": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "oavsr_op_gemm_chain": (C.c_int, [C.POINTER(GemmDesc), C.c_int32, _P, _P]),
    "oavsr_op_attention": (C.c_int, [C.POINTER(AttnDesc), _P]),
    "oavsr_op_decode_attention": (C.c_int, [C.POINTER(DecodeAttnDesc), _P]),
    "oavsr_op_norm": (C.c_int, [_I, _I, _P, _L, _P, _L, _P, _P, _I, _I, _F, _P]),
    "oavsr_op_rms_prep": (C.c_int, [_I, _P, _L, _P, _L, _P, _P, _I, _I, _P]),
    "oavsr_op_stem_pool": (C.c_int, [_I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "oavsr_op_conv3x3_c64": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "oavsr_op_conv3x3_c128": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "oavsr_op_beam_topk": (C.c_int, [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "oavsr_op_maxpool3x3s2": (C.c_int, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "oavsr_op_avgpool_hw": (C.c_int, [_I, _P, _P, _I, _I, _I, _P]),
    "oavsr_op_time_compress": (C.c_int, [_I, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "oavsr_op_copy_rows": (C.c_int, [_I, _P, _L, _P, _P, _L, _I, _I, _P]),
    "oavsr_op_rope_kv": (C.c_int, [_I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _I, _P]),
    "oavsr_op_swiglu": (C.c_int, [_I, _P, _L, _P, _L, _I, _I, _I, _P]),
    "oavsr_op_argmax": (C.c_int, [_P, _L, _I, _I, _I, _P, _P, _P]),
    "oavsr_op_build_rowinfo": (C.c_int, [_P, _P, _I, _I, _P, _I, _P]),
    "oavsr_op_logmel": (C.c_int, [_P, _I, _P, _L, _P, _I, _P, _P, _P, _L, _P]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C omni-avsr_amd/csrc`). omni_avsr_amd has no CPU / PyTorch fallback.")
    # torch first: its wheel bundles a HIP runtime, and the process must end up with ONE -- loaded the other way round (this library pulling in
    # /opt/rocm's libamdhip64 before torch brings its own) the second runtime finds "no ROCm-capable device"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.oavsr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libomniavsr_hip.so ABI version {lib.oavsr_abi_version()} != {ABI_VERSION} of this binding: rebuild the library (make -C omni-avsr_amd/csrc)")
    _lib = lib
    return lib


def env_report() -> dict:
    """OAVSR_* variables of this process: {name: value}; a trailing '?' on the value marks a name the library itself does not read."""
    buf = C.create_string_buffer(4096)
    load().oavsr_env_report(buf, len(buf))
    out = {}
    for item in buf.value.decode("utf-8", "replace").split(";"):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k] = v
    return out


def last_error() -> str:
    return load().oavsr_last_error().decode("utf-8", "replace")


class DeviceError(RuntimeError):
    """A HIP call inside the library failed (OAVSR_CHECK_HIP): the device / stream state is suspect, later calls will most likely fail
    the same way -- unlike an argument the library refused, which only concerns the call that carried it."""


def check(rc: int, after=None):
    """Raise for a non-zero return code of the library; `after` (optional) runs first either way (e.g. detach a per-call buffer)."""
    msg = last_error() if rc != 0 else ""
    if after is not None:
        after()
    if rc != 0:
        if msg.startswith("KeyError"):
            raise KeyError(msg[len("KeyError: "):])
        if " HIP error " in msg or rc == -2:
            raise DeviceError(f"libomniavsr_hip: {msg} (rc={rc})")
        raise RuntimeError(f"libomniavsr_hip: {msg} (rc={rc})")
