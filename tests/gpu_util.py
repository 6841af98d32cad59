"""Helpers for the -m gpu parity tests: call single kernels through the C ABI with torch-owned device memory."""
import ctypes as C

import torch

from omni_avsr_amd import lib as L

DEV = "cuda:0"


def tdt(dtype):
    return torch.bfloat16 if dtype == "bf16" else torch.float32


def dcode(dtype):
    return L.DT_BF16 if dtype == "bf16" else L.DT_F32


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tol(dtype, k=1):
    """abs tolerance relative to unit-scale outputs"""
    return 2e-2 if dtype == "bf16" else 1e-4


def gemm(dtype, A, W, *, A2=None, W2=None, scale=None, bias=None, prelu=None, act=0, residual=None, res_after_act=0,
         out_f32=False, kernel=0, M=None, N=None, K=None, C_out=None, collect=None, **geom):
    """One launch through oavsr_op_gemm.  collect: a list -- the descriptor is appended to it INSTEAD of being launched (oavsr_op_gemm_chain)."""
    lib = L.load()
    d = L.GemmDesc()
    d.dtype = dcode(dtype)
    d.M = M if M is not None else A.shape[0]
    d.N = N if N is not None else W.shape[0]
    d.K = K if K is not None else W.shape[1]
    d.A, d.W = P(A), P(W)
    d.lda = geom.pop("lda", A.stride(0) if A.dim() == 2 else 0)
    d.ldw = W.stride(0)
    if A2 is not None:
        d.A2, d.W2, d.K2, d.lda2, d.ldw2 = P(A2), P(W2), W2.shape[1], A2.stride(0), W2.stride(0)
    groups = geom.get("groups", 1)
    ncols = d.N * groups // (2 if geom.get("swiglu_pairs") else 1)
    out = C_out if C_out is not None else torch.empty(d.M, ncols, dtype=torch.float32 if out_f32 else tdt(dtype), device=DEV)
    d.C, d.ldc, d.out_f32 = P(out), out.stride(0), int(out_f32)
    d.col_scale, d.col_bias, d.prelu = P(scale), P(bias), P(prelu)
    d.act, d.res_after_act = act, res_after_act
    if residual is not None:
        d.residual, d.ldr = P(residual), residual.stride(0)
    d.groups = groups
    d.kernel = int(kernel)
    for k, v in geom.items():
        if k in ("rowinfo", "frameinfo", "partial", "sync", "row_sumsq", "ng_out", "ng_gain", "ng_rowsq"):
            setattr(d, k, P(v))
        elif k != "groups":
            setattr(d, k, v)
    if collect is not None:
        collect.append(d)
        return out
    L.check(lib.oavsr_op_gemm(C.byref(d), stream()))
    return out


def gemm_chain(descs, counters):
    """oavsr_op_gemm_chain: the collected launches as the phases of one launch.  counters: int32 tensor of len(descs) + 1 zeros."""
    arr = (L.GemmDesc * len(descs))(*descs)
    L.check(L.load().oavsr_op_gemm_chain(arr, len(descs), P(counters), stream()))


def digest_words(t):
    """The library's region digest (include/omni_avsr.h: oavsr_workspace_set_digests) of a tensor's bytes, computed on the host:
    sum_i splitmix64_finalizer(w_i | (i + 1) << 32) mod 2^64 over its 32-bit words; returned as a signed int64 like the device tensor holds it."""
    import numpy as np
    w = t.detach().contiguous().cpu().reshape(-1).view(torch.uint8).numpy().view(np.uint32).astype(np.uint64)
    z = w | ((np.arange(1, w.size + 1, dtype=np.uint64)) << np.uint64(32))
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
        tot = np.uint64(z.sum(dtype=np.uint64))
    return int(tot.astype(np.int64))
