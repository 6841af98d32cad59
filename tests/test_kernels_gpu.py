"""-m gpu: every HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of the same op.
Tolerances: f32 (exact-f32 MFMA) 1e-4 abs on unit-scale outputs; bf16 2e-2 (8-bit mantissa inputs)."""
import ctypes as C
import os
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omni_avsr_amd import lib as L  # noqa: E402
from gpu_util import DEV, P, dcode, gemm, stream, tdt, tol  # noqa: E402

DTYPES = ["f32", "bf16"]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x, dtype):
    """round to the compute dtype, return (device tensor in dtype, fp32 CPU view of the rounded values)"""
    xd = x.to(tdt(dtype))
    return xd.to(DEV).contiguous(), xd.float()


def close(out, ref, atol, rtol=0.0, what=""):
    out = out.float().cpu()
    err = (out - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    assert not bad.any(), f"{what}: max err {err.max().item():.3e} (limit {atol}), {int(bad.sum())} / {bad.numel()} bad, worst idx {int(err.argmax())}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 96, 256), (1, 64, 128), (333, 300, 384), (64, 1000, 128), (120, 65536 + 88, 256), (100, 33000, 128)])
def test_gemm_dense(dtype, M, N, K):
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    out = gemm(dtype, a, w)
    close(out, ar @ wr.T, tol(dtype), what="gemm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,K2,S", [(120, 2048, 8192, 0, 16), (120, 3072, 2048, 256, 4), (45, 192, 384, 0, 5), (120, 256, 2048, 0, 8),
                                        (8, 1024, 4096, 0, 2), (1, 64, 2048, 0, 8), (128, 1024 + 16, 1024, 128, 3), (97, 128, 512, 0, 5), (33, 2048, 2048, 0, 13)])
def test_gemm_splitk_reduced_inside_the_launch(dtype, M, N, K, K2, S):
    """Split-K of the decode-shaped GEMMs (M <= 128): fp32 partial slabs, summed in slab order by the tile's last-arriving workgroup inside
    the GEMM launch (gemm2.hip), with the fused epilogue (the deferred RMSNorm's row factor from partial sums of squares, bias, residual).  Against fp32 torch; repeated launches on the same
    counters and slabs give the same bits (the counters are zero again when a launch ends; the sum never depends on who came last)."""
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    D, eps = 256, 1e-5
    sq = (rnd(3, M, seed=8).abs() + 0.2) * D / 3          # three partial sums of squares per row -> 1 / rms per row
    rs = torch.rsqrt(sq.sum(0) / D + eps)
    rskw = dict(row_sumsq=sq.to(DEV).contiguous(), row_sumsq_tiles=3, row_dim=D, row_eps=eps)
    bias = rnd(N, seed=9)
    kw, ref = dict(rskw), ar @ wr.T
    if K2:
        a2, a2r = q(rnd(M, K2, seed=3), dtype)
        w2, w2r = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
        kw.update(A2=a2, W2=w2)
        ref = ref + a2r @ w2r.T
    part = torch.empty(S * M * N, dtype=torch.float32, device=DEV)
    sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV)
    out = gemm(dtype, a, w, residual=r, bias=bias.to(DEV), ksplit=S, partial=part, sync=sync, **kw).clone()
    close(out, ref * rs[:, None] + bias + rr, tol(dtype) * 2, rtol=(8e-3 if dtype == "bf16" else 1e-5), what="gemm split-K")
    for _ in range(3):
        again = gemm(dtype, a, w, residual=r, bias=bias.to(DEV), ksplit=S, partial=part, sync=sync, **kw)
        assert torch.equal(again, out), "repeated split-K launches differ"
    torch.cuda.synchronize()
    assert int(sync.abs().sum()) == 0, "the arrival counters are not zero after the launch"
    # the same reduce as a launch of its own (no arrival counters: what the engine runs above a few dozen rows): same code, same bits
    sep = gemm(dtype, a, w, residual=r, bias=bias.to(DEV), ksplit=S, partial=part, **kw)
    assert torch.equal(sep, out), "the stand-alone reduce launch differs from the reduce inside the GEMM launch"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,S", [(120, 2048, 8192, 16), (120, 2048, 2048, 8), (45, 4096, 1024, 4), (33, 3584, 512, 2), (8, 256, 256, 2), (128, 512, 384, 3)])
def test_gemm_splitk_next_norm_outputs(dtype, M, N, K, S):
    """The decode path's deferred RMSNorm (Llama_LoRA.py:587-655: residual add, then the next RMSNorm): the split-K reduce of o_proj /
    down_proj also writes T(x * gain) and, per 64-column tile, the rows' sums of squares for the consumer.  C is bit-identical to the launch without those outputs;
    they are checked against fp32 torch on the rounded C; in place on the residual rows, as the decoder uses it."""
    a, _ = q(rnd(M, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, _ = q(rnd(M, N, seed=5), dtype)
    gain = (rnd(N, seed=6) * 0.1 + 1).to(DEV)
    part = torch.empty(S * M * N, dtype=torch.float32, device=DEV)
    sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV)
    ref_c = gemm(dtype, a, w, residual=r, ksplit=S, partial=part, sync=sync).clone()
    rowsq = torch.empty(N // 64 * M, dtype=torch.float32, device=DEV)
    for _ in range(2):
        xg = torch.full((M, N), float("nan"), dtype=tdt(dtype), device=DEV)
        rowsq.fill_(float("nan"))
        out = gemm(dtype, a, w, residual=r, ksplit=S, partial=part, sync=sync, ng_out=xg, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq)
        assert torch.equal(out, ref_c), "C differs from the launch without the next-norm outputs"
        x = out.float()
        close(xg, (x * gain).cpu(), tol(dtype) if dtype == "bf16" else 1e-6, what="x * gain")
        sq = rowsq.view(-1, M)                              # [column tile][row]
        close(sq, (x * x).view(M, -1, 64).sum(-1).T.cpu(), 1e-3, rtol=1e-5, what="per-tile sums of squares")
        sq1 = sq.clone()
    r2 = r.clone()
    xg2 = torch.empty_like(xg)
    gemm(dtype, a, w, residual=r2, ksplit=S, partial=part, sync=sync, C_out=r2, ng_out=xg2, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq)
    assert torch.equal(r2, ref_c) and torch.equal(xg2, xg) and torch.equal(rowsq.view(-1, M), sq1)
    torch.cuda.synchronize()
    assert int(sync.abs().sum()) == 0
    # reduce as a launch of its own: same C, same T(x * gain), same sums of squares
    xg3, rowsq3 = torch.empty_like(xg), torch.full_like(rowsq, float("nan"))
    c3 = gemm(dtype, a, w, residual=r, ksplit=S, partial=part, ng_out=xg3, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq3)
    assert torch.equal(c3, ref_c) and torch.equal(xg3, xg) and torch.equal(rowsq3.view(-1, M), sq1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_decode_shaped_gemm_rows_do_not_depend_on_the_row_count(dtype):
    """The invariance the decode regime rests on: row r of a decode-shaped GEMM has the same bits whether the launch carries 1, 8, 33 or 128
    rows -- split (reduce inside the launch, next-norm outputs) and un-split (row_scale epilogue, fused SwiGLU, fp32 logits)."""
    N, K, S = 2048, 2048, 8
    a, _ = q(rnd(128, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, _ = q(rnd(128, N, seed=5), dtype)
    sq_row = ((rnd(128, seed=8).abs() + 0.5) * K).to(DEV)       # one sum of squares per row (as rms_prep leaves it)
    gain = (rnd(N, seed=6) * 0.1 + 1).to(DEV)
    part = torch.empty(S * 128 * N, dtype=torch.float32, device=DEV)
    sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV)
    rowsq = torch.empty(N // 64 * 128, dtype=torch.float32, device=DEV)
    wl, _ = q(rnd(4096 + 64, K, seed=3) / math.sqrt(K), dtype)          # "lm_head" / gate-up shaped: no split
    wr, _ = q(rnd(1000 + 5, K, seed=4) / math.sqrt(K), dtype)           # a ragged vocabulary (N % 16 != 0, odd row stride of the logits)
    full = {}
    for M in (128, 33, 17, 16, 8, 1):
        xg = torch.empty(M, N, dtype=tdt(dtype), device=DEV)
        rskw = dict(row_sumsq=sq_row[:M].contiguous(), row_sumsq_tiles=1, row_dim=K, row_eps=1e-5)
        c = gemm(dtype, a[:M], w, residual=r[:M].clone(), ksplit=S, partial=part, sync=sync, ng_out=xg, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq, **rskw)
        sq = rowsq[: N // 64 * M].view(-1, M).T.clone()     # [row][column tile]
        lg = gemm(dtype, a[:M], wl, out_f32=True, decode_tiled=1, **rskw)
        sg = gemm(dtype, a[:M], wl, swiglu_pairs=1, decode_tiled=1, **rskw)
        rg = gemm(dtype, a[:M], wr, out_f32=True, decode_tiled=1, **rskw)
        got = [t.clone() for t in (c, xg, sq, lg, sg, rg)]
        if M == 128:
            full = got
        else:
            for t, f, what in zip(got, full, ("C", "x * gain", "sums of squares", "fp32 logits", "SwiGLU", "ragged logits")):
                assert torch.equal(t, f[:M]), f"{dtype}: {what} of rows 0..{M - 1} depends on the row count ({M} vs 128)"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,K,K2,S", [(2048, 2048, 0, 8), (256, 2048, 0, 4), (3072, 2048, 256, 4), (2048, 8192, 0, 8), (512, 1024, 128, 5), (1024, 4096, 0, 16), (128, 512, 0, 3)])
def test_decode_strips_equal_the_tiled_kernels(dtype, N, K, K2, S):
    """<= 16 rows run on the weight-streaming strips (csrc/decode_strip.hip): a wave per (16-column strip, k-slice), the slices summed in slab order
    inside the workgroup -- no slabs, no arrival counters.  Against the SAME rows inside a 128-row launch of the tiled kernels (split-K over
    workgroups, reduce in a second launch): C, T(x * gain), the per-tile sums of squares rebuilt from the strip sums as (s0 + s1) + (s2 + s3),
    and a consumer that reads the strip layout (row_sumsq_form = 1) -- all bit-identical."""
    D = 4096
    a, _ = q(rnd(128, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, _ = q(rnd(128, N, seed=5), dtype)
    bias = rnd(N, seed=9).to(DEV)
    gain = (rnd(N, seed=6) * 0.1 + 1).to(DEV)
    kw = {}
    if K2:
        a2, _ = q(rnd(128, K2, seed=3), dtype)
        w2, _ = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
        kw.update(W2=w2)
    part = torch.empty(S * 128 * N, dtype=torch.float32, device=DEV)
    sq_in = ((rnd(128, seed=8).abs() + 0.5) * D).to(DEV)         # one sum of squares per row (rms_prep's layout)
    tiles = N // 64
    # the tiled kernels at 128 rows (reduce as its own launch)
    xg_t = torch.empty(128, N, dtype=tdt(dtype), device=DEV)
    rowsq_t = torch.empty(tiles * 128, dtype=torch.float32, device=DEV)
    c_t = gemm(dtype, a, w, residual=r.clone(), bias=bias, ksplit=S, partial=part, ng_out=xg_t, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq_t,
               row_sumsq=sq_in, row_sumsq_tiles=1, row_dim=D, row_eps=1e-5, **(dict(A2=a2, **kw) if K2 else {})).clone()
    rowsq_t = rowsq_t.view(tiles, 128)
    # a consumer of those sums on the tiled path: 128 rows, tile layout
    wc, _ = q(rnd(192, N, seed=12) / math.sqrt(N), dtype)
    cons_t = gemm(dtype, xg_t, wc, out_f32=True, decode_tiled=1, row_sumsq=rowsq_t.contiguous(), row_sumsq_tiles=tiles, row_dim=N, row_eps=1e-6).clone()
    for M in (16, 9, 8, 2, 1):
        xg = torch.full((M, N), float("nan"), dtype=tdt(dtype), device=DEV)
        rowsq = torch.full((N // 16 * M,), float("nan"), dtype=torch.float32, device=DEV)
        more = dict(A2=a2[:M], **kw) if K2 else {}
        c = gemm(dtype, a[:M], w, residual=r[:M].clone(), bias=bias, ksplit=S, ng_out=xg, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq, rowsq_strips=1,
                 row_sumsq=sq_in[:M].contiguous(), row_sumsq_tiles=1, row_dim=D, row_eps=1e-5, **more)
        assert torch.equal(c, c_t[:M]), f"{dtype} M={M}: C of the strips differs from the tiled kernels'"
        assert torch.equal(xg, xg_t[:M]), f"{dtype} M={M}: T(x * gain) differs"
        st = rowsq.view(N // 16, M)
        tile_sums = (st[0::4] + st[1::4]) + (st[2::4] + st[3::4])
        assert torch.equal(tile_sums, rowsq_t[:, :M]), f"{dtype} M={M}: (s0 + s1) + (s2 + s3) of the strip sums != the tiled producer's tile sums"
        cons = gemm(dtype, xg, wc, out_f32=True, decode_tiled=1, row_sumsq=rowsq, row_sumsq_tiles=tiles, row_dim=N, row_eps=1e-6, row_sumsq_form=1)
        assert torch.equal(cons, cons_t[:M]), f"{dtype} M={M}: a consumer of the strip layout differs from the tiled consumer"
        # ... and with the workspace's slab buffer and counters at hand, as the decoder launches it: few long strips (2048 x 8192: 128 strips of 256 KiB) are
        # then shared by PAIRS of workgroups -- the second continues the first's chain of additions (decode_strip.hip: PAIR) -- same bits, counters back at zero
        sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV)
        for rep in range(3):
            xg2 = torch.full((M, N), float("nan"), dtype=tdt(dtype), device=DEV)
            rowsq2 = torch.full((N // 16 * M,), float("nan"), dtype=torch.float32, device=DEV)
            c2 = gemm(dtype, a[:M], w, residual=r[:M].clone(), bias=bias, ksplit=S, partial=part, sync=sync, ng_out=xg2, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq2,
                      rowsq_strips=1, row_sumsq=sq_in[:M].contiguous(), row_sumsq_tiles=1, row_dim=D, row_eps=1e-5, **more)
            assert torch.equal(c2, c) and torch.equal(xg2, xg) and torch.equal(rowsq2, rowsq), f"{dtype} M={M}: the strips with slab buffer / counters at hand differ (launch {rep})"
        assert int(sync.abs().sum()) == 0
        # in place on the residual rows, as the decoder runs it
        r2 = r[:M].clone()
        gemm(dtype, a[:M], w, residual=r2, C_out=r2, bias=bias, ksplit=S, ng_out=xg, ng_ld=N, ng_gain=gain, ng_rowsq=rowsq, rowsq_strips=1,
             row_sumsq=sq_in[:M].contiguous(), row_sumsq_tiles=1, row_dim=D, row_eps=1e-5, **more)
        assert torch.equal(r2, c_t[:M])


def _decode_chain_case(dtype, M, d=2048, nq=2048, I=8192, k2=256, nqkv=3072, seed=0):
    """The five strip launches a decode step issues between two attention launches (model.hip: decode_layer_tail + the next layer's LoRA-down and
    QKV; widths of Llama-3.2-1B by default), on M rows: o_proj (+ residual, next-norm outputs) -> gate / up (+ SwiGLU, row factor) -> down_proj
    (+ residual, next-norm outputs) -> LoRA-down -> QKV (+ LoRA segment, row factor).  Returns (run, buffers): run(chain) re-initialises the
    state, issues the launches -- one by one or as ONE chained launch -- and returns copies of everything they wrote."""
    from gpu_util import gemm_chain
    t = tdt(dtype)
    W = dict(o=q(rnd(d, nq, seed=seed + 1) / math.sqrt(nq), dtype)[0], gu=q(rnd(2 * I, d, seed=seed + 2) / math.sqrt(d), dtype)[0],
             dn=q(rnd(d, I, seed=seed + 3) / math.sqrt(I), dtype)[0], la=q(rnd(k2, d, seed=seed + 4) / math.sqrt(d), dtype)[0],
             qkv=q(rnd(nqkv, d, seed=seed + 5) / math.sqrt(d), dtype)[0], lb=q(rnd(nqkv, k2, seed=seed + 6) / math.sqrt(k2), dtype)[0])
    attn0, _ = q(rnd(M, nq, seed=seed + 7), dtype)
    x0, _ = q(rnd(M, d, seed=seed + 8), dtype)
    g1 = (rnd(d, seed=seed + 9) * 0.1 + 1).to(DEV)
    g2 = (rnd(d, seed=seed + 10) * 0.1 + 1).to(DEV)
    bias = rnd(nqkv, seed=seed + 11).to(DEV)
    x = torch.empty(M, d, dtype=t, device=DEV)
    xg = torch.empty(M, d, dtype=t, device=DEV)
    rowsq = torch.empty(d // 16 * M, dtype=torch.float32, device=DEV)
    act = torch.empty(M, I, dtype=t, device=DEV)
    lo = torch.empty(M, k2, dtype=t, device=DEV)
    qkv = torch.empty(M, nqkv, dtype=t, device=DEV)
    outs = (x, xg, rowsq, act, lo, qkv)
    ctr = torch.zeros(8, dtype=torch.int32, device=DEV)

    def split(N, K, mn=1):      # model.hip: decode_split
        kt = K // (64 if dtype == "bf16" else 32)
        ks = min(256 // max(1, -(-N // 64)), max(1, kt // 8))
        return max(max(1, min(ks, 16)), min(mn, max(1, kt)))

    rs = dict(row_sumsq=rowsq, row_sumsq_tiles=d // 64, row_dim=d, row_eps=1e-5, row_sumsq_form=1)

    def run(chain):
        x.copy_(x0)
        for o in outs[1:]:
            o.fill_(float("nan"))
        c = [] if chain else None
        gemm(dtype, attn0, W["o"], residual=x, C_out=x, ksplit=split(d, nq, 2), ng_out=xg, ng_ld=d, ng_gain=g1, ng_rowsq=rowsq, rowsq_strips=1, collect=c)
        gemm(dtype, xg, W["gu"], swiglu_pairs=1, C_out=act, decode_tiled=1, collect=c, **rs)
        gemm(dtype, act, W["dn"], residual=x, C_out=x, ksplit=split(d, I, 2), ng_out=xg, ng_ld=d, ng_gain=g2, ng_rowsq=rowsq, rowsq_strips=1, collect=c)
        gemm(dtype, xg, W["la"], C_out=lo, ksplit=split(k2, d), collect=c)
        ks = split(nqkv, d + k2)
        gemm(dtype, xg, W["qkv"], A2=lo, W2=W["lb"], bias=bias, C_out=qkv, collect=c, **rs, **(dict(ksplit=ks) if ks > 1 else dict(decode_tiled=1)))
        if chain:
            gemm_chain(c, ctr)
        return [o.clone() for o in outs]

    return run, ctr


@pytest.mark.parametrize("dtype", DTYPES)
def test_decode_chain_equals_separate_launches(dtype):
    """oavsr_op_gemm_chain (csrc/decode_strip.hip: decode_chain_kernel): the strip launches between two attention launches of a decode step as the
    phases of ONE launch, each phase waiting -- weights already requested -- until the phase before has released its outputs.  Residual stream,
    T(x * gain), strip sums of squares, SwiGLU output, LoRA-down output and QKV: the bits of the five separate launches, at every ring depth
    (1 / 8 / 9 / 16 rows), launch after launch; the counters are back at zero and no wait was given up."""
    names = ("x", "T(x * gain)", "sums of squares", "SwiGLU", "LoRA-down", "QKV")
    for M in (1, 8, 9, 16):
        kw = dict(d=1024, nq=1024, I=4096, k2=128, nqkv=1536) if dtype == "f32" else {}
        run, ctr = _decode_chain_case(dtype, M, **kw)
        ref = run(False)
        assert not any(torch.isnan(o.float()).any() for o in ref)
        for rep in range(6):
            got = run(True)
            for n, a, b in zip(names, got, ref):
                assert torch.equal(a, b), f"{dtype} M={M} launch {rep}: {n} of the chained launch differs from the separate launches'"
        assert ctr.tolist() == [0] * 8, f"{dtype} M={M}: counters after the launches: {ctr.tolist()}"


def test_decode_chain_under_load_is_deterministic():
    """The chain's hand-off (release by every wave of the producer, agent-scope acquire behind the counter) with the chip shared: a second stream hammers
    HBM and a third runs other chains, so phases start late, spread over the XCDs and wait with their rings full."""
    run, ctr = _decode_chain_case("bf16", 8)
    run2, ctr2 = _decode_chain_case("bf16", 16, seed=50)
    ref, ref2 = run(False), run2(False)
    torch.cuda.synchronize()
    side, third = torch.cuda.Stream(), torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for rep in range(5):
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.add_(1)
        with torch.cuda.stream(third):
            others = [run2(True) for _ in range(3)]
        mine = [run(True) for _ in range(3)]
        torch.cuda.synchronize()
        for got in mine:
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), f"round {rep}: a chained launch under load differs"
        for got in others:
            assert all(torch.equal(a, b) for a, b in zip(got, ref2)), f"round {rep}: a chained launch on the third stream differs"
    assert ctr.tolist() == [0] * 8 and ctr2.tolist() == [0] * 8


def test_decode_strips_under_load_are_deterministic():
    """The strips' counted waits (wave-private LDS-DMA FIFO) only ever assume FEWER outstanding operations than there are, and a slot is refilled only after
    its fragments were read: with a second stream hammering HBM and a third one running the strips on other operands -- DMA arrivals late and out of step,
    other workgroups sharing the CUs -- every launch must return the bits of an undisturbed one.  All ring depths: 1 / 2 / 8 / 9 / 16 rows, short and long k-slices,
    and the PAIR form of the long strips (two workgroups per strip, the first half's sums handed over through memory)."""
    dtype = "bf16"
    shapes = [(2048, 2048, 0, 4), (3072, 2048, 256, 4), (2048, 8192, 0, 8), (16384, 2048, 0, 1), (256, 2048, 0, 4)]
    cases = []
    for N, K, K2, S in shapes:
        a, _ = q(rnd(16, K, seed=1), dtype)
        w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
        a2 = w2 = None
        if K2:
            a2, _ = q(rnd(16, K2, seed=3), dtype)
            w2, _ = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
        for M in (1, 2, 8, 9, 16):
            kw = dict(A2=a2[:M], W2=w2) if K2 else {}
            kw.update(dict(swiglu_pairs=1, decode_tiled=1) if S == 1 else dict(ksplit=S))
            cases.append((a[:M], w, kw, gemm(dtype, a[:M], w, **kw).clone()))
            if N * 16 <= 2048 * 16 and K >= 8192:      # few long strips: with a slab buffer and counters at hand, PAIRS of workgroups share a strip (hand-off through memory)
                pk = dict(kw, partial=torch.empty(S * M * N, dtype=torch.float32, device=DEV), sync=torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV))
                cases.append((a[:M], w, pk, cases[-1][3]))
    torch.cuda.synchronize()
    side, third = torch.cuda.Stream(), torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    wb, _ = q(rnd(4096, 4096, seed=9) / 64, dtype)
    ab, _ = q(rnd(16, 4096, seed=10), dtype)
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(30):
                junk.add_(1)
        with torch.cuda.stream(third):
            for _ in range(60):
                gemm(dtype, ab, wb, ksplit=4)
        for a_, w_, kw, ref in cases:
            out = gemm(dtype, a_, w_, **kw)
            assert torch.equal(out, ref), f"strips under load: {a_.shape[0]} x {w_.shape[0]} x {w_.shape[1]} differs from an undisturbed launch (repeat {rep})"
        torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", DTYPES)
def test_rms_prep_and_row_scale_equal_rmsnorm_then_gemm(dtype):
    """oavsr_op_rms_prep (the rows of a decode step's first layer: y = T(x * gain), the row's sum of squares) followed by a GEMM with the row
    factor computes RMSNorm(x) W^T (Llama_LoRA.py:587-655) -- against fp32 torch."""
    M, N, K = 24, 256, 512
    x, xr = q(rnd(M, K, seed=1) * 1.7, dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    gain = rnd(K, seed=3) * 0.1 + 1
    lib = L.load()
    y = torch.empty(M, K, dtype=tdt(dtype), device=DEV)
    sumsq = torch.empty(M, dtype=torch.float32, device=DEV)
    gd = gain.to(DEV)
    L.check(lib.oavsr_op_rms_prep(dcode(dtype), P(x), K, P(y), K, P(gd), P(sumsq), M, K, stream()))
    close(sumsq, xr.pow(2).sum(-1), 1e-2, rtol=1e-5, what="sum of squares")
    close(y, xr * gain, tol(dtype) if dtype == "bf16" else 1e-6, what="x * gain")
    out = gemm(dtype, y, w, row_sumsq=sumsq, row_sumsq_tiles=1, row_dim=K, row_eps=1e-5, decode_tiled=1)
    ref = (gain * xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)) @ wr.T
    close(out, ref, tol(dtype) * 2, what="deferred rmsnorm gemm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,K2", [(4096, 2048, 512, 0), (3000, 3072 + 40, 384, 128), (12000, 1024, 256, 0)])
def test_gemm_dma_pipeline(dtype, M, N, K, K2):
    """Shapes large enough to take the LDS-DMA 256x128 3-stage kernel (gemm2.hip): ragged M/N edges, 2nd K segment, epilogue."""
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    bias = rnd(N, seed=7)
    kw, ref = {}, ar @ wr.T
    if K2:
        a2, a2r = q(rnd(M, K2, seed=3), dtype)
        w2, w2r = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
        kw = dict(A2=a2, W2=w2)
        ref = ref + a2r @ w2r.T
    out = gemm(dtype, a, w, bias=bias.to(DEV), act=1, residual=r, res_after_act=1, **kw)
    close(out, F.gelu(ref + bias) + rr, tol(dtype) * 2, what="gemm dma")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dma_identity_layout(dtype):
    n = 2048
    a, _ = q(torch.eye(n), dtype)
    wv = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 253) / 64.0
    w, wr = q(wv, dtype)
    out = gemm(dtype, a, w)
    close(out, wr.T, 1e-6, what="gemm dma identity")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_dma(dtype):
    imgs, Cin, Cout, H, stride, ks = 420, 64, 128, 22, 2, 3
    x = rnd(imgs, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2) / math.sqrt(Cin * ks * ks)
    xd, xr = q(x.permute(0, 2, 3, 1).contiguous(), dtype)
    wd, wr = q(w.permute(0, 2, 3, 1).reshape(Cout, -1), dtype)
    OH = (H + 2 - ks) // stride + 1
    out = gemm(dtype, xd, wd, M=imgs * OH * OH, amode=1, H=H, W_=H, Cin=Cin, OH=OH, OW=OH, stride=stride, pad=1, ksize=ks, lda=0)
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.reshape(Cout, ks, ks, Cin).permute(0, 3, 1, 2), stride=stride, padding=1)
    close(out.reshape(imgs, OH, OH, Cout), ref.permute(0, 2, 3, 1), tol(dtype), what="conv2d dma")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride,Cin", [(1, 80), (2, 128)])
def test_conv1d_dma(dtype, stride, Cin):
    B, Tin, D = 4, 16000, 256
    Tout = Tin // stride
    x = rnd(B * Tin, Cin, seed=1)
    w = rnd(D, Cin, 3, seed=2) / math.sqrt(3 * Cin)
    b = rnd(D, seed=3)
    xd, xr = q(x, dtype)
    Kp = 256 if Cin == 80 else 3 * Cin
    wp = torch.zeros(D, Kp)
    wp[:, : 3 * Cin] = w.permute(0, 2, 1).reshape(D, -1)
    wd, wr = q(wp, dtype)
    ri = _rowinfo([i * Tout for i in range(B + 1)], [i * Tin for i in range(B + 1)], stride)
    out = gemm(dtype, xd, wd, M=B * Tout, N=D, K=Kp, amode=2, Cin=Cin, taps=3, Kvalid=3 * Cin, pad=1, rowinfo=ri, bias=b.to(DEV), act=1, lda=Cin)
    wref = wr[:, : 3 * Cin].reshape(D, 3, Cin).permute(0, 2, 1)
    ref = F.gelu(F.conv1d(xr.reshape(B, Tin, Cin).transpose(1, 2), wref, b, stride=stride, padding=1)).transpose(1, 2).reshape(B * Tout, D)
    close(out, ref, tol(dtype) * 2, what="conv1d dma")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(dtype):
    """A = I with an asymmetric W catches row/col swaps of the MFMA C layout."""
    n = 128
    a, _ = q(torch.eye(n), dtype)
    wv = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 / 64.0
    w, wr = q(wv, dtype)
    out = gemm(dtype, a, w)
    close(out, wr.T, 1e-6, what="gemm identity")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("res_after", [0, 1])
def test_gemm_epilogue(dtype, act, res_after):
    M, N, K = 150, 200, 128
    a, ar = q(rnd(M, K, seed=3), dtype)
    w, wr = q(rnd(N, K, seed=4) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    scale = (rnd(N, seed=6) * 0.1 + 1).to(DEV)
    bias = rnd(N, seed=7).to(DEV)
    slope = (rnd(N, seed=8) * 0.05 + 0.25).to(DEV)
    out = gemm(dtype, a, w, scale=scale, bias=bias, prelu=slope if act == 3 else None, act=act, residual=r, res_after_act=res_after)
    y = (ar @ wr.T) * scale.cpu() + bias.cpu()
    if not res_after:
        y = y + rr
    y = {0: lambda v: v, 1: F.gelu, 2: F.relu, 3: lambda v: torch.where(v >= 0, v, slope.cpu() * v), 4: F.silu}[act](y)
    if res_after:
        y = y + rr
    close(out, y, tol(dtype) * 2, what="gemm epilogue")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("decode", [False, True])
def test_gemm_lora_segment(dtype, decode):
    """C = A1 W1^T + A2 W2^T (fused W + BA), fp32 logits output."""
    M, N, K, K2 = (8 if decode else 100), 320, 256, 64
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    a2, a2r = q(rnd(M, K2, seed=3), dtype)
    w2, w2r = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
    bias = rnd(N, seed=9).to(DEV)
    out = gemm(dtype, a, w, A2=a2, W2=w2, bias=bias, out_f32=True, decode_tiled=int(decode))
    close(out, ar @ wr.T + a2r @ w2r.T + bias.cpu(), tol(dtype), what="gemm 2-segment")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cin,Cout,H,stride,ks", [(64, 64, 22, 1, 3), (64, 128, 22, 2, 3), (64, 128, 22, 2, 1), (128, 256, 11, 2, 3), (256, 512, 6, 2, 3), (512, 512, 3, 1, 3)])
def test_conv2d_implicit_gemm(dtype, Cin, Cout, H, stride, ks):
    imgs = 5
    x = rnd(imgs, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2) / math.sqrt(Cin * ks * ks)
    xd, xr = q(x.permute(0, 2, 3, 1).contiguous(), dtype)            # NHWC
    wd, wr = q(w.permute(0, 2, 3, 1).reshape(Cout, -1), dtype)
    pad = ks // 2
    OH = (H + 2 * pad - ks) // stride + 1
    out = gemm(dtype, xd, wd, M=imgs * OH * OH, amode=1, H=H, W_=H, Cin=Cin, OH=OH, OW=OH, stride=stride, pad=pad, ksize=ks, lda=0)
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.reshape(Cout, ks, ks, Cin).permute(0, 3, 1, 2), stride=stride, padding=pad)
    close(out.reshape(imgs, OH, OH, Cout), ref.permute(0, 2, 3, 1), tol(dtype), what="conv2d")


@pytest.mark.parametrize("imgs,H,with_res,with_prelu", [(7, 22, True, True), (300, 22, False, True), (3, 13, True, False), (1, 22, False, False)])
def test_conv3x3_c64_image_resident(imgs, H, with_res, with_prelu):
    """conv.hip: image + weights resident in LDS; must equal the implicit-GEMM path bit for bit (same k order, same epilogue)
    and match torch conv2d + BN + residual + PReLU; residual aliasing the output is the trunk's in-place update."""
    dtype = "bf16"
    x = rnd(imgs, 64, H, H, seed=1)
    w = rnd(64, 64, 3, 3, seed=2) / math.sqrt(576)
    xd, xr = q(x.permute(0, 2, 3, 1).contiguous(), dtype)
    wd, wr = q(w.permute(0, 2, 3, 1).reshape(64, -1), dtype)
    scale, shift, slope = rnd(64, seed=3) * 0.1 + 1, rnd(64, seed=4) * 0.1, rnd(64, seed=5) * 0.05 + 0.25
    res, resr = q(rnd(imgs * H * H, 64, seed=6), dtype)
    sc, sh, sl = scale.to(DEV), shift.to(DEV), slope.to(DEV)
    kw = dict(scale=sc, bias=sh)
    if with_prelu:
        kw.update(prelu=sl, act=3)
    if with_res:
        kw.update(residual=res)
    want = gemm(dtype, xd, wd, M=imgs * H * H, amode=1, H=H, W_=H, Cin=64, OH=H, OW=H, stride=1, pad=1, ksize=3, lda=0, **kw)
    out = res.clone() if with_res else torch.empty(imgs * H * H, 64, dtype=tdt(dtype), device=DEV)      # residual aliases the output
    L.check(L.load().oavsr_op_conv3x3_c64(P(xd), imgs, H, H, P(wd), P(sc), P(sh), P(sl) if with_prelu else None,
                                          P(out) if with_res else None, P(out), stream()))
    assert torch.equal(out, want), f"direct conv differs from implicit GEMM: max {(out.float() - want.float()).abs().max().item():.3e}"
    y = F.conv2d(xr.permute(0, 3, 1, 2), wr.reshape(64, 3, 3, 64).permute(0, 3, 1, 2), padding=1)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1).reshape(imgs * H * H, 64)
    if with_res:
        y = y + resr
    if with_prelu:
        y = torch.where(y >= 0, y, slope.view(1, -1) * y)
    close(out, y, tol(dtype) * 2, what="direct conv")


@pytest.mark.parametrize("imgs,H,with_res,with_prelu", [(7, 11, True, True), (601, 11, False, True), (2, 11, True, False), (1, 11, False, False), (5, 8, True, True), (33, 6, True, True)])
def test_conv3x3_c128_image_resident(imgs, H, with_res, with_prelu):
    """conv.hip, 128 channels (trunk layer2, resnet.py:61-74): groups of images resident in LDS, the weights streamed tap by tap; must equal the
    implicit-GEMM path bit for bit (same k order, same epilogue) and match torch conv2d + BN + residual + PReLU; odd image counts (a last partial
    group), several groups per workgroup (601 images on 256 CUs), residual aliasing the output as in the trunk's in-place update."""
    dtype, Cn = "bf16", 128
    x = rnd(imgs, Cn, H, H, seed=1)
    w = rnd(Cn, Cn, 3, 3, seed=2) / math.sqrt(9 * Cn)
    xd, xr = q(x.permute(0, 2, 3, 1).contiguous(), dtype)
    wd, wr = q(w.permute(0, 2, 3, 1).reshape(Cn, -1), dtype)
    scale, shift, slope = rnd(Cn, seed=3) * 0.1 + 1, rnd(Cn, seed=4) * 0.1, rnd(Cn, seed=5) * 0.05 + 0.25
    res, resr = q(rnd(imgs * H * H, Cn, seed=6), dtype)
    sc, sh, sl = scale.to(DEV), shift.to(DEV), slope.to(DEV)
    kw = dict(scale=sc, bias=sh)
    if with_prelu:
        kw.update(prelu=sl, act=3)
    if with_res:
        kw.update(residual=res)
    import os
    assert os.environ.get("OAVSR_CONV_C128", "1") != "0"
    # the implicit GEMM of the same convolution (Cin = 127 + 1 would change the math; the GEMM entry point never takes the image-resident kernel)
    want = gemm(dtype, xd, wd, M=imgs * H * H, amode=1, H=H, W_=H, Cin=Cn, OH=H, OW=H, stride=1, pad=1, ksize=3, lda=0, **kw)
    out = res.clone() if with_res else torch.empty(imgs * H * H, Cn, dtype=tdt(dtype), device=DEV)      # residual aliases the output
    L.check(L.load().oavsr_op_conv3x3_c128(P(xd), imgs, H, H, P(wd), P(sc), P(sh), P(sl) if with_prelu else None,
                                           P(out) if with_res else None, P(out), stream()))
    assert torch.equal(out, want), f"image-resident conv differs from implicit GEMM: max {(out.float() - want.float()).abs().max().item():.3e}"
    y = F.conv2d(xr.permute(0, 3, 1, 2), wr.reshape(Cn, 3, 3, Cn).permute(0, 3, 1, 2), padding=1)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1).reshape(imgs * H * H, Cn)
    if with_res:
        y = y + resr
    if with_prelu:
        y = torch.where(y >= 0, y, slope.view(1, -1) * y)
    close(out, y, tol(dtype) * 2, what="image-resident conv, 128 channels")


def _rowinfo(cu_out, cu_in, stride):
    lib = L.load()
    n = len(cu_out) - 1
    rows = cu_out[-1]
    co = torch.tensor(cu_out, dtype=torch.int32, device=DEV)
    ci = torch.tensor(cu_in, dtype=torch.int32, device=DEV)
    ri = torch.empty(rows, 4, dtype=torch.int32, device=DEV)
    L.check(lib.oavsr_op_build_rowinfo(P(co), P(ci), n, stride, P(ri), rows, stream()))
    return ri


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D,G,taps,lens", [(128, 4, 16, [37, 5, 64]), (256, 4, 32, [200, 61, 129, 3])])
def test_conv1d_posconv_grouped_varlen(dtype, D, G, taps, lens):
    """AV-HuBERT positional conv: groups, even kernel with SamePad, GELU, residual, per-utterance zero halo.  The second shape has AV-HuBERT-
    Large's 64 channels per group and >= 256 rows: the grouped 128 x 64 LDS-DMA conv1d tile (round 5)."""
    Cg = D // G
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    x = rnd(cu[-1], D, seed=1)
    w = rnd(D, Cg, taps, seed=2) / math.sqrt(Cg * taps)
    b = rnd(D, seed=3)
    xd, xr = q(x, dtype)
    wd, wr = q(w.permute(0, 2, 1).reshape(D, -1), dtype)
    ri = _rowinfo(cu, cu, 1)
    out = gemm(dtype, xd, wd, M=cu[-1], N=Cg, K=taps * Cg, amode=2, Cin=Cg, taps=taps, Kvalid=taps * Cg, pad=taps // 2, groups=G, a_goff=Cg,
               rowinfo=ri, bias=b.to(DEV), act=1, residual=xd, res_after_act=1, lda=D)
    refs = []
    wref = wr.reshape(D, taps, Cg).permute(0, 2, 1)
    for i, n in enumerate(lens):
        xi = xr[cu[i]:cu[i + 1]].T[None]
        y = F.conv1d(xi, wref, b, padding=taps // 2, groups=G)[:, :, :-1]
        refs.append((xi + F.gelu(y))[0].T)
    close(out, torch.cat(refs), tol(dtype) * 2, what="posconv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride,Cin", [(1, 80), (2, 128)])
def test_conv1d_whisper(dtype, stride, Cin):
    B, Tin, D = 2, 40, 64
    Tout = Tin // stride
    x = rnd(B * Tin, Cin, seed=1)
    w = rnd(D, Cin, 3, seed=2) / math.sqrt(3 * Cin)
    b = rnd(D, seed=3)
    xd, xr = q(x, dtype)
    Kp = 256 if Cin == 80 else 3 * Cin
    wp = torch.zeros(D, Kp)
    wp[:, : 3 * Cin] = w.permute(0, 2, 1).reshape(D, -1)
    wd, wr = q(wp, dtype)
    cu_in = [i * Tin for i in range(B + 1)]
    cu_out = [i * Tout for i in range(B + 1)]
    ri = _rowinfo(cu_out, cu_in, stride)
    out = gemm(dtype, xd, wd, M=B * Tout, N=D, K=Kp, amode=2, Cin=Cin, taps=3, Kvalid=3 * Cin, pad=1, rowinfo=ri, bias=b.to(DEV), act=1, lda=Cin)
    wref = wr[:, : 3 * Cin].reshape(D, 3, Cin).permute(0, 2, 1)
    ref = F.gelu(F.conv1d(xr.reshape(B, Tin, Cin).transpose(1, 2), wref, b, stride=stride, padding=1)).transpose(1, 2).reshape(B * Tout, D)
    close(out, ref, tol(dtype) * 2, what="whisper conv")


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_conv3d(dtype):
    """5x7x7 stride (1,2,2) conv on f32 video with per-utterance temporal zero padding + BN affine + PReLU."""
    lens = [3, 4]
    Fr = sum(lens)
    vid = rnd(Fr, 88, 88, seed=1)
    w = rnd(64, 1, 5, 7, 7, seed=2) / math.sqrt(245)
    scale = (rnd(64, seed=3) * 0.1 + 1)
    shift = rnd(64, seed=4) * 0.1
    slope = rnd(64, seed=5) * 0.05 + 0.25
    wp = torch.zeros(64, 256)
    wp[:, :245] = w.reshape(64, 245)
    wd, wr = q(wp, dtype)
    fi = torch.tensor([[t, n] for n in lens for t in range(n)], dtype=torch.int32, device=DEV)
    out = gemm(dtype, vid.to(DEV), wd, M=Fr * 1936, N=64, K=256, amode=3, frameinfo=fi, scale=scale.to(DEV), bias=shift.to(DEV), prelu=slope.to(DEV), act=3, lda=0)
    refs = []
    o = 0
    vq = vid.to(tdt(dtype)).float()          # the loader rounds the video to the compute dtype
    for n in lens:
        v = vq[o:o + n][None, None]
        y = F.conv3d(v, wr[:, :245].reshape(64, 1, 5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3))
        y = y * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
        y = torch.where(y >= 0, y, slope.view(1, -1, 1, 1, 1) * y)
        refs.append(y[0].permute(1, 2, 3, 0).reshape(n * 1936, 64))
        o += n
    close(out, torch.cat(refs), tol(dtype) * 2, what="stem")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("lens", [[3, 4], [1], [6, 2, 9]])
def test_stem_conv_pool_fused(dtype, lens):
    """Fused stem (stem.hip): conv3d + BN + PReLU + MaxPool3d((1,3,3), s (1,2,2), p (0,1,1)) vs torch, incl. utterance borders."""
    Fr = sum(lens)
    vid = rnd(Fr, 88, 88, seed=1)
    w = rnd(64, 1, 5, 7, 7, seed=2) / math.sqrt(245)
    scale = (rnd(64, seed=3) * 0.1 + 1)
    shift = rnd(64, seed=4) * 0.1
    slope = rnd(64, seed=5) * 0.05 + 0.25
    wp = torch.zeros(64, 256)
    wp[:, :245] = w.reshape(64, 245)
    wd, wr = q(wp, dtype)
    fi = torch.tensor([[t, n] for n in lens for t in range(n)], dtype=torch.int32, device=DEV)
    vd = vid.to(DEV)
    scratch = torch.empty(64 * 288, dtype=tdt(dtype), device=DEV)
    out = torch.empty(Fr, 22, 22, 64, dtype=tdt(dtype), device=DEV)
    sc, sh, sl = scale.to(DEV), shift.to(DEV), slope.to(DEV)
    L.check(L.load().oavsr_op_stem_pool(dcode(dtype), P(vd), 0, 88, 88, P(fi), Fr, P(wd), P(scratch), P(sc), P(sh), P(sl), P(out), stream()))
    refs = []
    o = 0
    vq = vid.to(tdt(dtype)).float()
    for n in lens:
        v = vq[o:o + n][None, None]
        y = F.conv3d(v, wr[:, :245].reshape(64, 1, 5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3))
        y = y * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
        y = torch.where(y >= 0, y, slope.view(1, -1, 1, 1, 1) * y)
        y = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        refs.append(y[0].permute(1, 2, 3, 0))
        o += n
    close(out.reshape(Fr * 484, 64), torch.cat(refs).reshape(Fr * 484, 64), tol(dtype) * 2, what="fused stem")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W", [(3, 96, 96), (1, 96, 96), (3, 97, 101), (1, 88, 88)])
def test_stem_uint8_frames_with_fused_test_transform(dtype, C, H, W):
    """uint8 frame source of the fused stem: x / 255, CenterCrop(88), Grayscale, Normalize(0.421, 0.165) (datamodule/transforms.py:93-99)
    inside the window gather.  Bar: the SAME BITS as the host transform (harness.video_test_transform, fp32 torch ops) followed by the
    f32-source kernel -- the pixel arithmetic is replayed operation by operation, so nothing may differ."""
    from omni_avsr_amd.harness import video_test_transform
    lens = [5, 3]
    Fr = sum(lens)
    g = torch.Generator().manual_seed(C * 1000 + H)
    frames = torch.randint(0, 256, (Fr, C, H, W), generator=g, dtype=torch.uint8)
    w = rnd(64, 1, 5, 7, 7, seed=2) / math.sqrt(245)
    wp = torch.zeros(64, 256)
    wp[:, :245] = w.reshape(64, 245)
    wd, _ = q(wp, dtype)
    sc, sh, sl = (rnd(64, seed=3) * 0.1 + 1).to(DEV), (rnd(64, seed=4) * 0.1).to(DEV), (rnd(64, seed=5) * 0.05 + 0.25).to(DEV)
    fi = torch.tensor([[t, n] for n in lens for t in range(n)], dtype=torch.int32, device=DEV)
    scratch = torch.empty(64 * 288, dtype=tdt(dtype), device=DEV)
    outs = []
    host = video_test_transform(frames)[:, 0].contiguous()          # [F, 88, 88] f32, what the reference's dataset hands to forward()
    for src, fmt in ((host.to(DEV), (0, 88, 88)), (frames.to(DEV), (C, H, W))):
        out = torch.empty(Fr, 22, 22, 64, dtype=tdt(dtype), device=DEV)
        L.check(L.load().oavsr_op_stem_pool(dcode(dtype), P(src), *fmt, P(fi), Fr, P(wd), P(scratch), P(sc), P(sh), P(sl), P(out), stream()))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), f"uint8 source differs from host transform + f32 source: max |d| = {(outs[0].float() - outs[1].float()).abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qr, kr, vr, cu_q, cu_k, H, KV, hd, scale, causal):
    outs = []
    for s in range(len(cu_q) - 1):
        qs = qr[cu_q[s]:cu_q[s + 1]].reshape(-1, H, hd).transpose(0, 1)
        ks = kr[cu_k[s]:cu_k[s + 1]].reshape(-1, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        vs = vr[cu_k[s]:cu_k[s + 1]].reshape(-1, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        w = qs @ ks.transpose(1, 2) * scale
        if causal:
            ql, kl = qs.shape[1], ks.shape[1]
            mask = torch.arange(kl)[None, :] > (torch.arange(ql)[:, None] + (kl - ql))
            w = w.masked_fill(mask, float("-inf"))
        o = torch.softmax(w, -1) @ vs
        outs.append(o.transpose(0, 1).reshape(-1, H * hd))
    return torch.cat(outs)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd,H,KV", [(64, 4, 4), (64, 8, 2), (128, 4, 1)])
@pytest.mark.parametrize("causal", [0, 1])
def test_attention_varlen(dtype, hd, H, KV, causal):
    lens = [70, 1, 130, 64]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    rows = cu[-1]
    nq, nkv = H * hd, KV * hd
    qkv = rnd(rows, nq + 2 * nkv, seed=11)
    qkvd, qkvr = q(qkv, dtype)
    out = torch.empty(rows, nq, dtype=tdt(dtype), device=DEV)
    cud = torch.tensor(cu, dtype=torch.int32, device=DEV)
    d = L.AttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode(dtype), hd, H, KV, len(lens), max(lens), causal
    esz = qkvd.element_size()
    d.Q, d.K, d.V, d.O = qkvd.data_ptr(), qkvd.data_ptr() + nq * esz, qkvd.data_ptr() + (nq + nkv) * esz, out.data_ptr()
    d.ldq = d.ldk = d.ldv = nq + 2 * nkv
    d.ldo = nq
    d.cu_q, d.cu_k, d.scale = cud.data_ptr(), cud.data_ptr(), 1.0 / math.sqrt(hd)
    L.check(L.load().oavsr_op_attention(C.byref(d), stream()))
    ref = _attn_ref(qkvr[:, :nq], qkvr[:, nq:nq + nkv], qkvr[:, nq + nkv:], cu, cu, H, KV, hd, 1.0 / math.sqrt(hd), causal)
    close(out, ref, tol(dtype), what="attention")


@pytest.mark.parametrize("H,KV,causal,lens", [(4, 4, 0, [1500, 1500]), (8, 2, 0, [70, 1, 130, 64, 333]), (4, 4, 1, [200, 129, 31]), (2, 1, 1, [1, 300])])
def test_attention_32x32_form(H, KV, causal, lens):
    """attn_dma32_kernel (form 32: 32x32x16 MFMAs, one tile of 32 queries per wave; what the Whisper encoder runs): against the fp32
    reference like the 16x16 kernels -- ragged lengths, GQA, the causal edge tiles -- and within rounding of the default form."""
    hd = 64
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    rows, nq, nkv = cu[-1], H * hd, KV * hd
    qkvd, qkvr = q(rnd(rows, nq + 2 * nkv, seed=21), "bf16")
    cud = torch.tensor(cu, dtype=torch.int32, device=DEV)
    outs = {}
    for form in (0, 32):
        out = torch.full((rows, nq), float("nan"), dtype=torch.bfloat16, device=DEV)
        d = L.AttnDesc()
        d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode("bf16"), hd, H, KV, len(lens), max(lens), causal
        d.Q, d.K, d.V, d.O = qkvd.data_ptr(), qkvd.data_ptr() + nq * 2, qkvd.data_ptr() + (nq + nkv) * 2, out.data_ptr()
        d.ldq = d.ldk = d.ldv = nq + 2 * nkv
        d.ldo = nq
        d.cu_q, d.cu_k, d.scale, d.form = cud.data_ptr(), cud.data_ptr(), 1.0 / math.sqrt(hd), form
        L.check(L.load().oavsr_op_attention(C.byref(d), stream()))
        outs[form] = out
    ref = _attn_ref(qkvr[:, :nq], qkvr[:, nq:nq + nkv], qkvr[:, nq + nkv:], cu, cu, H, KV, hd, 1.0 / math.sqrt(hd), causal)
    close(outs[32], ref, tol("bf16"), what="attention, 32x32 form")
    close(outs[32], outs[0].float().cpu(), tol("bf16"), what="attention, 32x32 form vs default form")
    again = outs[32].clone()
    L.check(L.load().oavsr_op_attention(C.byref(d), stream()))
    assert torch.equal(outs[32], again)


def test_attention_short_sequences_after_nan_poisoned_lds():
    """The bf16 / head_dim 64 kernel relies on the buffer range check for key rows past a sequence's end (they must land in LDS as
    zeros: a masked probability of 0 times a stale NaN would poison the output).  Poison the LDS of every CU with NaN patterns first
    (a big GEMM on NaN inputs stages them through all of its LDS), then run many short ragged sequences (1024 workgroups, so every CU
    takes several)."""
    hd, H, KV = 64, 8, 8
    g = torch.Generator().manual_seed(3)
    lens = [int(x) for x in torch.randint(1, 150, (128,), generator=g)]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    rows, nq, nkv = cu[-1], H * hd, KV * hd
    qkvd, qkvr = q(rnd(rows, nq + 2 * nkv, seed=12), "bf16")
    out = torch.empty(rows, nq, dtype=torch.bfloat16, device=DEV)
    cud = torch.tensor(cu, dtype=torch.int32, device=DEV)
    d = L.AttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode("bf16"), hd, H, KV, len(lens), max(lens), 0
    d.Q, d.K, d.V, d.O = qkvd.data_ptr(), qkvd.data_ptr() + nq * 2, qkvd.data_ptr() + (nq + nkv) * 2, out.data_ptr()
    d.ldq = d.ldk = d.ldv = nq + 2 * nkv
    d.ldo = nq
    d.cu_q, d.cu_k, d.scale = cud.data_ptr(), cud.data_ptr(), 1.0 / math.sqrt(hd)
    nan = torch.full((4096, 1024), float("nan"), dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        gemm("bf16", nan, nan)                       # 256 x 256 tiles on every CU: both operands are NaN everywhere in LDS
        L.check(L.load().oavsr_op_attention(C.byref(d), stream()))
        assert torch.isfinite(out.float()).all(), "key rows past the end of a sequence were not zero-filled"
    ref = _attn_ref(qkvr[:, :nq], qkvr[:, nq:nq + nkv], qkvr[:, nq + nkv:], cu, cu, H, KV, hd, 1.0 / math.sqrt(hd), 0)
    close(out, ref, tol("bf16"), what="attention after NaN-poisoned LDS")


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_softmax_spike(dtype):
    """One key dominates late in the sequence: exercises the online-softmax rescale branch."""
    hd, H, n = 64, 2, 200
    qv = rnd(n, H * hd, seed=1)
    kv = rnd(n, H * hd, seed=2)
    vv = rnd(n, H * hd, seed=3)
    kv[150] = qv[10] * 3.0
    qkvd, qkvr = q(torch.cat([qv, kv, vv], 1), dtype)
    out = torch.empty(n, H * hd, dtype=tdt(dtype), device=DEV)
    cud = torch.tensor([0, n], dtype=torch.int32, device=DEV)
    d = L.AttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode(dtype), hd, H, H, 1, n, 0
    esz = qkvd.element_size()
    d.Q, d.K, d.V, d.O = qkvd.data_ptr(), qkvd.data_ptr() + H * hd * esz, qkvd.data_ptr() + 2 * H * hd * esz, out.data_ptr()
    d.ldq = d.ldk = d.ldv = 3 * H * hd
    d.ldo = H * hd
    d.cu_q, d.cu_k, d.scale = cud.data_ptr(), cud.data_ptr(), 1.0 / math.sqrt(hd)
    L.check(L.load().oavsr_op_attention(C.byref(d), stream()))
    ref = _attn_ref(qkvr[:, :H * hd], qkvr[:, H * hd:2 * H * hd], qkvr[:, 2 * H * hd:], [0, n], [0, n], H, H, hd, 1.0 / math.sqrt(hd), 0)
    close(out, ref, tol(dtype), what="attention spike")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd,H,KV", [(64, 8, 2), (128, 7, 1)])
def test_decode_attention(dtype, hd, H, KV):
    utts, beams, maxp, maxt = 3, 1, 96, 8
    rows = utts * beams
    plen = [50, 96, 1]
    step = 5
    qd, qr = q(rnd(rows, H * hd, seed=1), dtype)
    kp, kpr = q(rnd(utts, KV, maxp, hd, seed=2), dtype)
    vp, vpr = q(rnd(utts, KV, maxp, hd, seed=3), dtype)
    kt, ktr = q(rnd(rows, KV, maxt, hd, seed=4), dtype)
    vt, vtr = q(rnd(rows, KV, maxt, hd, seed=5), dtype)
    out = torch.empty(rows, H * hd, dtype=tdt(dtype), device=DEV)
    pl = torch.tensor(plen, dtype=torch.int32, device=DEV)
    st = torch.tensor([step], dtype=torch.int32, device=DEV)
    d = L.DecodeAttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.rows, d.beams, d.max_prefix, d.max_tail = dcode(dtype), hd, H, KV, rows, beams, maxp, maxt
    d.Q, d.Kp, d.Vp, d.Kt, d.Vt, d.O = qd.data_ptr(), kp.data_ptr(), vp.data_ptr(), kt.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.ldq = d.ldo = H * hd
    d.prefix_len, d.step, d.anc, d.scale = pl.data_ptr(), st.data_ptr(), None, 1.0 / math.sqrt(hd)
    L.check(L.load().oavsr_op_decode_attention(C.byref(d), stream()))
    refs = []
    G = H // KV
    for r in range(rows):
        k = torch.cat([kpr[r, :, :plen[r]], ktr[r, :, :step]], 1).repeat_interleave(G, 0)      # [H,ctx,hd]
        v = torch.cat([vpr[r, :, :plen[r]], vtr[r, :, :step]], 1).repeat_interleave(G, 0)
        qq = qr[r].reshape(H, 1, hd)
        w = torch.softmax(qq @ k.transpose(1, 2) / math.sqrt(hd), -1)
        refs.append((w @ v).reshape(H * hd))
    close(out, torch.stack(refs), tol(dtype), what="decode attention")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd,H,KV", [(64, 8, 2), (128, 6, 2)])
def test_decode_attention_fused_rope_append(dtype, hd, H, KV):
    """RoPE on q / new k + append of k,v to the tail cache inside the decode-attention kernel, long context (> 4 tiles)."""
    from omni_avsr_amd.weights import rope_inv_freq
    utts, maxp, maxt = 2, 320, 8
    plen, step = [300, 77], 3                      # tail holds 2 old tokens, the kernel appends the 3rd
    G = H // KV
    ld = (H + 2 * KV) * hd
    qkv, qkvr = q(rnd(utts, ld, seed=1), dtype)
    kp, kpr = q(rnd(utts, KV, maxp, hd, seed=2), dtype)
    vp, vpr = q(rnd(utts, KV, maxp, hd, seed=3), dtype)
    kt, ktr = q(rnd(utts, KV, maxt, hd, seed=4), dtype)
    vt, vtr = q(rnd(utts, KV, maxt, hd, seed=5), dtype)
    inv = rope_inv_freq(hd, 500000.0, None)
    pos = torch.tensor([p + step - 1 for p in plen], dtype=torch.int32)
    invd, posd = inv.to(DEV), pos.to(DEV)
    cposd = torch.full((utts,), step - 1, dtype=torch.int32, device=DEV)
    pl = torch.tensor(plen, dtype=torch.int32, device=DEV)
    st = torch.tensor([step], dtype=torch.int32, device=DEV)
    out = torch.empty(utts, H * hd, dtype=tdt(dtype), device=DEV)
    d = L.DecodeAttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.rows, d.beams, d.max_prefix, d.max_tail = dcode(dtype), hd, H, KV, utts, 1, maxp, maxt
    d.Q, d.Kp, d.Vp, d.Kt, d.Vt, d.O = qkv.data_ptr(), kp.data_ptr(), vp.data_ptr(), kt.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.ldq, d.ldo = ld, H * hd
    d.prefix_len, d.step, d.anc, d.scale = pl.data_ptr(), st.data_ptr(), None, 1.0 / math.sqrt(hd)
    d.inv_freq, d.pos, d.cpos = invd.data_ptr(), posd.data_ptr(), cposd.data_ptr()
    L.check(L.load().oavsr_op_decode_attention(C.byref(d), stream()))
    ang = pos.float()[:, None] * inv[None]
    cos, sin = torch.cat([ang, ang], -1).cos()[:, None], torch.cat([ang, ang], -1).sin()[:, None]
    rot = lambda x: x * cos + torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1) * sin
    qh = rot(qkvr[:, :H * hd].reshape(utts, H, hd)).to(tdt(dtype)).float()
    kn = rot(qkvr[:, H * hd:(H + KV) * hd].reshape(utts, KV, hd)).to(tdt(dtype)).float()
    vn = qkvr[:, (H + KV) * hd:].reshape(utts, KV, hd)
    refs = []
    for r in range(utts):
        k = torch.cat([kpr[r, :, :plen[r]], ktr[r, :, :step - 1], kn[r][:, None]], 1).repeat_interleave(G, 0)
        v = torch.cat([vpr[r, :, :plen[r]], vtr[r, :, :step - 1], vn[r][:, None]], 1).repeat_interleave(G, 0)
        w = torch.softmax(qh[r][:, None] @ k.transpose(1, 2) / math.sqrt(hd), -1)
        refs.append((w @ v).reshape(H * hd))
    close(out, torch.stack(refs), tol(dtype), what="decode attention fused rope")
    close(kt[:, :, step - 1], kn, tol(dtype), what="appended k")
    close(vt[:, :, step - 1], vn, 0.0, what="appended v")


@pytest.mark.parametrize("hd,H,KV,maxp", [(64, 32, 8, 320), (128, 32, 8, 320), (64, 32, 8, 1216)])
def test_decode_attention_wave_count_follows_the_rows_not_the_bits(hd, H, KV, maxp):
    """Greedy bf16 rows: a launch of few rows (rows x kv heads <= the CU count) spreads a row's 64-key tiles over EIGHT waves, a full step over two
    (csrc/attn.hip: launch_decode_attention).  A row's keys are accumulated per SLOT (tile mod 8) and the slots folded in one canonical order --
    in registers by two waves, from LDS by eight -- so a row's output has the same bits in both: rows 0 .. 5 of a 70-row launch against a 6-row
    launch, with the fused RoPE + append, contexts of 2 .. 6 tiles and (max_prefix 1216) of more than eight tiles, where a slot holds two."""
    from omni_avsr_amd.weights import rope_inv_freq
    dtype, R, maxt, step = "bf16", 70, 8, 4
    ld = (H + 2 * KV) * hd
    g = torch.Generator().manual_seed(11)
    plen = [int(x) for x in torch.randint(70, maxp - 4, (R,), generator=g)]
    plen[:6] = [maxp - 7, 65, 129, 313, 200, maxp - 64]
    qkv, _ = q(rnd(R, ld, seed=1), dtype)
    kp, _ = q(rnd(R, KV, maxp, hd, seed=2), dtype)
    vp, _ = q(rnd(R, KV, maxp, hd, seed=3), dtype)
    inv = rope_inv_freq(hd, 500000.0, None).to(DEV)
    outs = []
    for rows in (R, 6):
        kt, _ = q(rnd(R, KV, maxt, hd, seed=4), dtype)
        vt, _ = q(rnd(R, KV, maxt, hd, seed=5), dtype)
        pos = torch.tensor([p_ + step - 1 for p_ in plen[:rows]], dtype=torch.int32, device=DEV)
        cpos = torch.full((rows,), step - 1, dtype=torch.int32, device=DEV)
        pl = torch.tensor(plen[:rows], dtype=torch.int32, device=DEV)
        st = torch.tensor([step], dtype=torch.int32, device=DEV)
        out = torch.empty(rows, H * hd, dtype=tdt(dtype), device=DEV)
        d = L.DecodeAttnDesc()
        d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.rows, d.beams, d.max_prefix, d.max_tail = dcode(dtype), hd, H, KV, rows, 1, maxp, maxt
        d.Q, d.Kp, d.Vp, d.Kt, d.Vt, d.O = qkv.data_ptr(), kp.data_ptr(), vp.data_ptr(), kt.data_ptr(), vt.data_ptr(), out.data_ptr()
        d.ldq, d.ldo = ld, H * hd
        d.prefix_len, d.step, d.anc, d.scale = pl.data_ptr(), st.data_ptr(), None, 1.0 / math.sqrt(hd)
        d.inv_freq, d.pos, d.cpos = inv.data_ptr(), pos.data_ptr(), cpos.data_ptr()
        L.check(L.load().oavsr_op_decode_attention(C.byref(d), stream()))
        outs.append((out.clone(), kt[:rows, :, step - 1].clone(), vt[:rows, :, step - 1].clone()))
    big, small = outs
    assert torch.isfinite(big[0].float()).all()
    for what, a, b in zip(("O", "appended k", "appended v"), big, small):
        assert torch.equal(a[:6], b), f"head_dim {hd}: {what} of a row depends on how many rows share the launch"


@pytest.mark.parametrize("beams,fused", [(15, True), (15, False), (6, True), (4, False)])
def test_decode_attention_beam_groups_of_four(beams, fused):
    """decode_attn_beam4_kernel (bf16, head_dim 64, GQA group 4, beams > 1): four beams of an utterance per workgroup share the prefix K / V
    tiles; every beam's tail comes from ITS ancestry-addressed rows (another beam's tail must never leak into its columns), the last group
    is ragged (15 = 4 + 4 + 4 + 3), the fused RoPE + append writes every beam's new k / v to its own tail row."""
    from omni_avsr_amd.weights import rope_inv_freq
    dtype, hd, H, KV = "bf16", 64, 8, 2
    G = H // KV
    utts, maxp, maxt = 3, 320, 32
    rows = utts * beams
    plen, step = [313, 64, 7], 6                  # tail: 5 old tokens (+ the appended 6th when fused)
    ld = (H + 2 * KV) * hd
    qkv, qkvr = q(rnd(rows, ld, seed=1), dtype)
    kp, kpr = q(rnd(utts, KV, maxp, hd, seed=2), dtype)
    vp, vpr = q(rnd(utts, KV, maxp, hd, seed=3), dtype)
    kt, ktr = q(rnd(rows, KV, maxt, hd, seed=4), dtype)
    vt, vtr = q(rnd(rows, KV, maxt, hd, seed=5), dtype)
    g = torch.Generator().manual_seed(9)
    anc = torch.empty(rows, maxt, dtype=torch.int32)
    for u in range(utts):                          # position t of beam b lives in the tail row of a random beam of the same utterance
        anc[u * beams:(u + 1) * beams] = u * beams + torch.randint(0, beams, (beams, maxt), generator=g, dtype=torch.int32)
    nold = step - 1 if fused else step
    if fused:
        anc[:, step - 1] = torch.arange(rows, dtype=torch.int32)     # the appended token lands in the beam's own row
    inv = rope_inv_freq(hd, 500000.0, None)
    pos = torch.tensor([plen[r // beams] + step - 1 for r in range(rows)], dtype=torch.int32)
    pl = torch.tensor(plen, dtype=torch.int32, device=DEV)
    st = torch.tensor([step], dtype=torch.int32, device=DEV)
    ancd, invd, posd = anc.to(DEV), inv.to(DEV), pos.to(DEV)
    cposd = torch.full((rows,), step - 1, dtype=torch.int32, device=DEV)
    out = torch.full((rows, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
    d = L.DecodeAttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.rows, d.beams, d.max_prefix, d.max_tail = dcode(dtype), hd, H, KV, rows, beams, maxp, maxt
    d.Q, d.Kp, d.Vp, d.Kt, d.Vt, d.O = qkv.data_ptr(), kp.data_ptr(), vp.data_ptr(), kt.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.ldq, d.ldo = (ld if fused else H * hd), H * hd
    if not fused:
        qonly = qkv[:, :H * hd].contiguous()
        d.Q = qonly.data_ptr()
    d.prefix_len, d.step, d.anc, d.scale = pl.data_ptr(), st.data_ptr(), ancd.data_ptr(), 1.0 / math.sqrt(hd)
    if fused:
        d.inv_freq, d.pos, d.cpos = invd.data_ptr(), posd.data_ptr(), cposd.data_ptr()
    L.check(L.load().oavsr_op_decode_attention(C.byref(d), stream()))
    if fused:
        ang = pos.float()[:, None] * inv[None]
        cos, sin = torch.cat([ang, ang], -1).cos()[:, None], torch.cat([ang, ang], -1).sin()[:, None]
        rot = lambda x: x * cos + torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1) * sin
        qh = rot(qkvr[:, :H * hd].reshape(rows, H, hd)).to(torch.bfloat16).float()
        kn = rot(qkvr[:, H * hd:(H + KV) * hd].reshape(rows, KV, hd)).to(torch.bfloat16).float()
        vn = qkvr[:, (H + KV) * hd:].reshape(rows, KV, hd)
    else:
        qh = qkvr[:, :H * hd].reshape(rows, H, hd)
    refs = []
    for r in range(rows):
        u = r // beams
        kts = torch.stack([ktr[int(anc[r, t]), :, t] for t in range(nold)], 1)      # [KV, nold, hd]
        vts = torch.stack([vtr[int(anc[r, t]), :, t] for t in range(nold)], 1)
        ks, vs = [kpr[u, :, :plen[u]], kts], [vpr[u, :, :plen[u]], vts]
        if fused:
            ks.append(kn[r][:, None]); vs.append(vn[r][:, None])
        k = torch.cat(ks, 1).repeat_interleave(G, 0)
        v = torch.cat(vs, 1).repeat_interleave(G, 0)
        w = torch.softmax(qh[r][:, None] @ k.transpose(1, 2) / math.sqrt(hd), -1)
        refs.append((w @ v).reshape(H * hd))
    close(out, torch.stack(refs), tol(dtype), what="beam-grouped decode attention")
    if fused:
        close(kt[:, :, step - 1], kn, tol(dtype), what="appended k")
        close(vt[:, :, step - 1], vn, 0.0, what="appended v")


# ------------------------------------------------------------------------------------------------ norms & elementwise
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("D", [128, 1024, 3584])
def test_norms(dtype, mode, D):
    rows = 37
    xd, xr = q(rnd(rows, D, seed=1) * 2 + 0.3, dtype)
    Dw = 2 * D if mode == 2 else D
    w = (rnd(Dw, seed=2) * 0.1 + 1).to(DEV)
    b = (rnd(Dw, seed=3) * 0.1).to(DEV)
    y = torch.empty(rows, Dw, dtype=tdt(dtype), device=DEV)
    eps = 1e-5
    L.check(L.load().oavsr_op_norm(mode, dcode(dtype), P(xd), D, P(y), Dw, P(w), P(b), rows, D, eps, stream()))
    if mode == 0:
        ref = F.layer_norm(xr, (D,), w.cpu(), b.cpu(), eps)
    elif mode == 1:
        ref = w.cpu() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps))
    else:
        ref = F.layer_norm(torch.cat([torch.zeros_like(xr), xr], -1), (2 * D,), w.cpu(), b.cpu(), eps)
    close(y, ref, tol(dtype) * 2, what=f"norm mode {mode}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pools(dtype):
    imgs, H, Cc = 3, 44, 64
    xd, xr = q(rnd(imgs, H, H, Cc, seed=1), dtype)
    y = torch.empty(imgs, 22, 22, Cc, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_maxpool3x3s2(dcode(dtype), P(xd), P(y), imgs, H, H, Cc, stream()))
    ref = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    close(y, ref, 1e-6, what="maxpool")
    x2d, x2r = q(rnd(imgs, 9, 512, seed=2), dtype)
    y2 = torch.empty(imgs, 512, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_avgpool_hw(dcode(dtype), P(x2d), P(y2), imgs, 9, 512, stream()))
    close(y2, x2r.mean(1), tol(dtype), what="avgpool")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stack", [0, 1])
@pytest.mark.parametrize("rate", [1, 2, 5])
def test_time_compress(dtype, stack, rate):
    D = 128
    lens = [13, 25, 4]
    offs = [0, 20, 50]              # sequences with gaps in the input buffer (Whisper: 1500-token slots, keep < 1500)
    xd, xr = q(rnd(60, D, seed=1), dtype)
    nout = [n // rate for n in lens]
    seg = torch.tensor([[offs[i], sum(nout[:i]), nout[i], 0] for i in range(3)], dtype=torch.int32, device=DEV)
    Do = D * rate if stack else D
    y = torch.zeros(sum(nout), Do, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_time_compress(dcode(dtype), P(xd), D, P(y), Do, P(seg), 3, max(nout), rate, D, stack, stream()))
    refs = []
    for i, n in enumerate(lens):
        blk = xr[offs[i]:offs[i] + nout[i] * rate].reshape(nout[i], rate, D)
        refs.append(blk.reshape(nout[i], rate * D) if stack else blk.mean(1))
    close(y, torch.cat(refs), tol(dtype), what="time compress")


@pytest.mark.parametrize("dtype", DTYPES)
def test_copy_rows_and_swiglu_and_argmax(dtype):
    lib = L.load()
    D = 256
    td, tr = q(rnd(50, D, seed=1), dtype)
    mp = torch.tensor([3, -1, 49, 0, 7], dtype=torch.int32, device=DEV)
    y = torch.full((5, D), 7.0, dtype=tdt(dtype), device=DEV)
    L.check(lib.oavsr_op_copy_rows(dcode(dtype), P(td), D, P(mp), P(y), D, 5, D, stream()))
    ref = torch.stack([tr[3], torch.full((D,), 7.0), tr[49], tr[0], tr[7]])
    close(y, ref, 0.0, what="copy_rows")
    I = 384
    xd, xr = q(rnd(9, 2 * I, seed=2), dtype)
    y2 = torch.empty(9, I, dtype=tdt(dtype), device=DEV)
    L.check(lib.oavsr_op_swiglu(dcode(dtype), P(xd), 2 * I, P(y2), I, 9, I, 0, stream()))
    close(y2, F.silu(xr[:, :I]) * xr[:, I:], tol(dtype), what="swiglu")
    L.check(lib.oavsr_op_swiglu(dcode(dtype), P(xd), 2 * I, P(y2), I, 9, I, 1, stream()))
    xi = xr.reshape(9, I // 16, 2, 16)
    close(y2, (F.silu(xi[:, :, 0]) * xi[:, :, 1]).reshape(9, I), tol(dtype), what="swiglu interleaved")
    lg = rnd(6, 5000, seed=3).to(DEV)
    lg[2, 100] = lg[2, 4000] = 50.0     # tie -> lowest index
    lg[4, 77] = 99.0
    lgc = lg.cpu().clone()
    lgc[:, 77] = float("-inf")
    scratch = torch.empty(6 * 64 * 2, dtype=torch.float32, device=DEV)
    for scr in (None, scratch):             # single-stage and two-stage variants
        o = torch.full((6,), -1, dtype=torch.int32, device=DEV)
        L.check(lib.oavsr_op_argmax(P(lg), 5000, 6, 5000, 77, P(o), P(scr), stream()))
        assert o.cpu().tolist() == lgc.argmax(-1).tolist()
        assert o[2].item() == 100


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [64, 128])
def test_rope_kv(dtype, hd):
    from omni_avsr_amd.weights import rope_inv_freq
    H, KV, rows, maxc = 4, 2, 10, 16
    ld = (H + 2 * KV) * hd
    xd, xr = q(rnd(rows, ld, seed=1), dtype)
    inv = rope_inv_freq(hd, 500000.0, dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192))
    pos = torch.tensor([0, 1, 2, 3, 4, 300, 301, 7, 8, 9], dtype=torch.int32)
    slot = torch.tensor([0, 0, 0, 0, 0, 1, 1, 2, 2, 2], dtype=torch.int32)
    cpos = torch.tensor([0, 1, 2, 3, 4, 0, 1, 5, 6, 7], dtype=torch.int32)
    kc = torch.zeros(3, KV, maxc, hd, dtype=tdt(dtype), device=DEV)
    vc = torch.zeros_like(kc)
    invd, posd, slotd, cposd = inv.to(DEV), pos.to(DEV), slot.to(DEV), cpos.to(DEV)     # keep the device buffers alive
    L.check(L.load().oavsr_op_rope_kv(dcode(dtype), rows, H, KV, hd, P(xd), ld, P(invd), P(posd), P(slotd), P(cposd), P(kc), P(vc), maxc, stream()))
    ang = pos.float()[:, None] * inv[None]
    cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
    qk = xr[:, :(H + KV) * hd].reshape(rows, H + KV, hd)
    rot = torch.cat([-qk[..., hd // 2:], qk[..., :hd // 2]], -1)
    ref = xr.clone()
    ref[:, :(H + KV) * hd] = (qk * cos[:, None] + rot * sin[:, None]).reshape(rows, -1)
    close(xd, ref, tol(dtype), what="rope")
    kref = torch.zeros(3, KV, maxc, hd)
    vref = torch.zeros(3, KV, maxc, hd)
    refq = ref.to(tdt(dtype)).float()
    for r in range(rows):
        kref[slot[r], :, cpos[r]] = refq[r, H * hd:(H + KV) * hd].reshape(KV, hd)
        vref[slot[r], :, cpos[r]] = refq[r, (H + KV) * hd:].reshape(KV, hd)
    close(kc, kref, tol(dtype), what="k cache")
    close(vc, vref, tol(dtype), what="v cache")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,I,K", [(2504, 8192, 256), (1000, 4096, 128), (4100, 1024 + 32, 192)])
def test_gemm_dma_fused_swiglu_prefill(dtype, M, I, K):
    """SwiGLU in the LDS-DMA GEMM epilogue (prefill): gate / up rows interleaved by 16; must equal GEMM -> swiglu kernel bit for
    bit (both operands are rounded to the storage type first) and match the fp32 reference."""
    x, xr = q(rnd(M, K, seed=1), dtype)
    wg, wgr = q(rnd(I, K, seed=2) / math.sqrt(K), dtype)
    wu, wur = q(rnd(I, K, seed=3) / math.sqrt(K), dtype)
    inter = torch.stack([wg.reshape(-1, 16, K), wu.reshape(-1, 16, K)], 1).reshape(2 * I, K).contiguous()
    fused = gemm(dtype, x, inter, swiglu_pairs=1)
    assert fused.shape == (M, I)
    gu = gemm(dtype, x, inter)
    two = torch.empty(M, I, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_swiglu(dcode(dtype), P(gu), 2 * I, P(two), I, M, I, 1, stream()))
    assert torch.equal(fused, two), f"fused epilogue differs from gemm + swiglu: {(fused.float() - two.float()).abs().max().item():.3e}"
    g_ = (xr @ wgr.T).to(tdt(dtype)).float()
    u_ = (xr @ wur.T).to(tdt(dtype)).float()
    close(fused, F.silu(g_) * u_, tol(dtype) * 2, rtol=(8e-3 if dtype == "bf16" else 0.0), what="fused swiglu (prefill)")


# ------------------------------------------------------------------------------------------------ beam-search candidates
@pytest.mark.parametrize("rows,V,K,mode", [(120, 128256, 15, "random"), (7, 128256, 15, "clustered"), (5, 151936, 4, "ties"), (3, 1000, 32, "random"),
                                           (2, 40, 15, "random"), (4, 128261, 15, "flat"), (3, 128261, 15, "plateau")])
def test_beam_topk_fast_and_fallback(rows, V, K, mode):
    """Top-2K of log_softmax(logits) + beam score per row.  "clustered" puts the whole top-2K on indices that ONE thread of the fast kernel owns
    (stride 1024: its threshold comes from the other threads' maxima, the candidates are still complete); "ties" checks the lower-index-first
    rule; "flat" (all logits equal) and "plateau" (2 000 elements tied at the 2K-th value) overflow the candidate list -> exact fallback."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(rows, V, generator=g) * 3
    if mode == "clustered":
        for r in range(rows):
            x[r, (torch.arange(40) * 1024 + 5 + r) % V] = 20.0 + torch.arange(40).float()
    if mode == "ties":
        x = torch.round(x * 2) / 2               # many exact ties
    if mode == "flat":
        x = torch.zeros(rows, V)
    if mode == "plateau":
        for r in range(rows):
            x[r, torch.randperm(V, generator=g)[:2000]] = 15.0
            x[r, torch.randperm(V, generator=g)[:7]] = 30.0 + r
    bs = torch.randn(rows, generator=g)
    xd, bsd = x.to(DEV), bs.to(DEV)
    twoK = min(2 * K, V)
    outs = []
    for use_fast in (True, False):
        val = torch.full((rows, 2 * K), float("nan"), device=DEV)
        tok = torch.full((rows, 2 * K), -1, dtype=torch.int32, device=DEV)
        flag = torch.zeros(rows, dtype=torch.int32, device=DEV)
        L.check(L.load().oavsr_op_beam_topk(P(xd), V, rows, V, K, P(bsd), P(val), P(tok), P(flag) if use_fast else None, stream()))
        outs.append((val.cpu(), tok.cpu(), flag.cpu()))
    (vf, tf, ff), (ve, te, _) = outs
    assert torch.equal(tf[:, :twoK], te[:, :twoK]) and torch.equal(vf[:, :twoK], ve[:, :twoK]), "fast path differs from the exact kernel"
    if mode in ("flat", "plateau"):
        assert int(ff.sum()) == rows, "rows with more ties at the 2K-th value than the candidate list holds must have taken the fallback"
    if mode in ("random", "clustered", "ties") and V > 100000:
        assert int(ff.sum()) == 0, "these rows should be served by the fast path"
    lp = torch.log_softmax(x, -1) + bs[:, None]
    order = torch.argsort(torch.stack([-x, torch.arange(V).expand(rows, V).float()], -1)[..., 0], dim=-1, stable=True)[:, :twoK]
    assert torch.equal(te[:, :twoK].long(), order), "candidate tokens differ from a stable descending sort"
    close(ve[:, :twoK], torch.gather(lp, 1, order), 1e-4, what="candidate scores")


# ------------------------------------------------------------------------------------------------ 8-wave ping-pong GEMM (gemm3.hip)
@pytest.mark.parametrize("M,N,K,K2", [(4096, 4096, 1024, 0), (12000, 3072, 256, 0), (5000, 3072 + 40, 512, 128), (3001, 8192, 64, 64)])
def test_gemm_8p_pipeline(M, N, K, K2):
    """>= 144 tiles of 256 x 256 -> the 8-wave ping-pong kernel: ragged M / N edges, a second K segment, 1..16 k-tiles (the DMA runs two
    k-tiles ahead and fills its tail with dummy loads), fused epilogue."""
    dtype = "bf16"
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    bias = rnd(N, seed=7)
    kw, ref = {}, ar @ wr.T
    if K2:
        a2, a2r = q(rnd(M, K2, seed=3), dtype)
        w2, w2r = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
        kw = dict(A2=a2, W2=w2)
        ref = ref + a2r @ w2r.T
    out = gemm(dtype, a, w, bias=bias.to(DEV), act=1, residual=r, res_after_act=1, kernel=2, **kw)
    close(out, F.gelu(ref + bias) + rr, tol(dtype) * 2, what="gemm 8p")
    out2 = gemm(dtype, a, w, bias=bias.to(DEV), act=1, residual=r, res_after_act=1, kernel=2, **kw)
    assert torch.equal(out, out2), "two launches of the ping-pong GEMM differ (race in the k-loop schedule)"


def test_gemm_8p_identity_layout():
    """A = I with an asymmetric W: catches a swapped quadrant / fragment map of the four-quadrant wave tile."""
    n = 4096
    a, _ = q(torch.eye(n), "bf16")
    wv = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 253) / 64.0
    w, wr = q(wv, "bf16")
    out = gemm("bf16", a, w, kernel=2)
    close(out, wr.T, 1e-6, what="gemm 8p identity")


def test_gemm_8p_bitwise_equals_16_wave_kernel():
    """Same accumulation order per output element as the 16-wave 256 x 256 kernel (kernel selectors 2 / 5 of oavsr_op_gemm): the two must
    agree bit for bit, so a tensor has the same bits whichever kernel the dispatcher picked."""
    g = torch.Generator().manual_seed(11)
    a = torch.randn(6000, 1024, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(4096, 1024, generator=g) / 32).to(DEV, torch.bfloat16)
    a2 = torch.randn(6000, 128, generator=g).to(DEV, torch.bfloat16)
    w2 = (torch.randn(4096, 128, generator=g) / 11).to(DEV, torch.bfloat16)
    b = torch.randn(4096, generator=g).to(DEV)
    outs = [gemm("bf16", a, w, A2=a2, W2=w2, bias=b, act=1, kernel=sel) for sel in (2, 5)]
    assert torch.equal(outs[0], outs[1]), f"ping-pong vs 16-wave kernel: max diff {(outs[0].float() - outs[1].float()).abs().max().item():.3e}"


def test_gemm_8p_fused_swiglu_equals_gemm_plus_swiglu_kernel():
    dtype, M, I, K = "bf16", 2504, 4096, 192
    x, xr = q(rnd(M, K, seed=1), dtype)
    wg, _ = q(rnd(I, K, seed=2) / math.sqrt(K), dtype)
    wu, _ = q(rnd(I, K, seed=3) / math.sqrt(K), dtype)
    inter = torch.stack([wg.reshape(-1, 16, K), wu.reshape(-1, 16, K)], 1).reshape(2 * I, K).contiguous()
    fused = gemm(dtype, x, inter, swiglu_pairs=1, kernel=2)
    gu = gemm(dtype, x, inter)
    two = torch.empty(M, I, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_swiglu(dcode(dtype), P(gu), 2 * I, P(two), I, M, I, 1, stream()))
    assert torch.equal(fused, two), f"ping-pong fused SwiGLU differs from gemm + swiglu: {(fused.float() - two.float()).abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------ persistent ping-pong GEMM (gemm4.hip)
@pytest.mark.parametrize("M,N,K,grid,act", [
    (2048, 1024, 192, 8, 1),          # 32 full tiles on 8 workgroups: 4 rounds, the shortest k loop (3 k-tiles: first, second-to-last, last)
    (2048, 1024, 256, 8, 1),          # 4 k-tiles: one steady k-tile between them
    (2048 + 77, 1280 + 40, 320, 16, 1),   # ragged M and N (edge tiles take the conservative waits), 5 k-tiles: the buffer parity flips per tile
    (3000, 2048, 1024, 24, 0),        # 96 tiles on 24 workgroups; no activation
    (4096, 2304, 512, 40, 4),         # 144 tiles / 40: a partial last round (24 tiles dealt over the 8 XCD groups)
    (12000, 3072, 1024, 0, 0),        # Whisper QKV at the benchmark's size, the launcher's own grid (192 workgroups, 3 rounds)
])
def test_gemm_8pp_persistent(M, N, K, grid, act):
    """Persistent ping-pong kernel: one workgroup walks several 256 x 256 tiles, the LDS-DMA stream continues across tiles and the C stores
    are waited for by count only.  Against fp32 torch; bit-identical to the 16-wave kernel and to the non-persistent ping-pong kernel (same
    accumulation order and epilogue expressions); repeated launches identical (a race in the cross-tile schedule would show as a diff)."""
    dtype = "bf16"
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    bias = rnd(N, seed=7)
    ref = ar @ wr.T + bias
    ref = F.gelu(ref) if act == 1 else F.silu(ref) if act == 4 else ref
    out = gemm(dtype, a, w, bias=bias.to(DEV), act=act, kernel=3, grid=grid)
    close(out, ref, tol(dtype) * 2, what="gemm 8pp")
    for sel in (5, 2):
        other = gemm(dtype, a, w, bias=bias.to(DEV), act=act, kernel=sel)
        assert torch.equal(out, other), f"persistent kernel vs kernel selector {sel}: max diff {(out.float() - other.float()).abs().max().item():.3e}"
    for _ in range(3):
        again = gemm(dtype, a, w, bias=bias.to(DEV), act=act, kernel=3, grid=grid)
        assert torch.equal(out, again), "two launches of the persistent GEMM differ (race in the cross-tile schedule)"
    if act in (0, 1):                 # the 192-row form (second A half-tile of 64 rows; instantiated for no activation and bias + GELU): same bits
        for _ in range(2):
            rows192 = gemm(dtype, a, w, bias=bias.to(DEV), act=act, kernel=8, grid=grid)
            assert torch.equal(out, rows192), f"192-row form differs: max diff {(out.float() - rows192.float()).abs().max().item():.3e}"


@pytest.mark.parametrize("M,N,K,grid,act,raa", [
    (12000, 1024, 4096, 0, 0, 0),       # Whisper fc2: 188 tiles, one per workgroup (ragged single round dealt over the XCD groups), residual
    (2048 + 77, 1280 + 40, 320, 16, 1, 1),   # ragged edges, several rounds, GELU then residual
    (3000, 1024, 1024, 8, 4, 0),        # 48 tiles on 8 workgroups: the residual variant across tile boundaries
    (3000 + 50, 1024, 1024, 8, 0, 0),   # no activation, ragged M, several rounds: also the 192-row form across tile boundaries
    (12000, 1024, 1024, 0, 0, 1),       # Whisper o_proj
])
def test_gemm_8pp_residual_variant(M, N, K, grid, act, raa):
    """The residual form of the persistent kernel (o_proj / fc2 / down_proj: residual and activation after the transposition, bias before
    it): vs fp32 torch, bit-identical to the 16-wave and the ping-pong kernel, in place (C aliases the residual, as the encoders call it)."""
    dtype = "bf16"
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    bias = rnd(N, seed=7)
    ref = ar @ wr.T + bias
    if raa:
        ref = (F.gelu(ref) if act == 1 else F.silu(ref) if act == 4 else ref) + rr
    else:
        ref = ref + rr
        ref = F.gelu(ref) if act == 1 else F.silu(ref) if act == 4 else ref
    out = gemm(dtype, a, w, bias=bias.to(DEV), act=act, residual=r, res_after_act=raa, kernel=3, grid=grid)
    close(out, ref, tol(dtype) * 2, what="gemm 8pp residual")
    for sel in (5, 2):
        other = gemm(dtype, a, w, bias=bias.to(DEV), act=act, residual=r, res_after_act=raa, kernel=sel)
        assert torch.equal(out, other), f"persistent residual variant vs kernel selector {sel}: max diff {(out.float() - other.float()).abs().max().item():.3e}"
    inplace = r.clone()
    gemm(dtype, a, w, bias=bias.to(DEV), act=act, residual=inplace, res_after_act=raa, kernel=3, grid=grid, C_out=inplace)
    assert torch.equal(inplace, out), "in-place (C == residual) differs"
    if act == 0:                      # 192-row form of the residual variant (Whisper fc2 / o_proj: 252 tiles instead of 188)
        rows192 = gemm(dtype, a, w, bias=bias.to(DEV), act=act, residual=r, res_after_act=raa, kernel=8, grid=grid)
        assert torch.equal(out, rows192), f"192-row residual form differs: max diff {(out.float() - rows192.float()).abs().max().item():.3e}"
        inplace = r.clone()
        gemm(dtype, a, w, bias=bias.to(DEV), act=act, residual=inplace, res_after_act=raa, kernel=8, grid=grid, C_out=inplace)
        assert torch.equal(inplace, out), "192-row form in place (C == residual) differs"


def test_gemm_8pp_identity_layout_and_tile_walk():
    """A = I with an asymmetric W on 8 workgroups: every output element names its (row, column), so a tile written to the wrong place, a
    skipped tile or a swapped quadrant of any round shows."""
    n = 2048
    a, _ = q(torch.eye(n), "bf16")
    wv = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 253) / 64.0
    w, wr = q(wv, "bf16")
    for grid in (8, 16, 48, 64):
        out = gemm("bf16", a, w, kernel=3, grid=grid)
        close(out, wr.T, 1e-6, what=f"gemm 8pp identity grid {grid}")
        out = gemm("bf16", a, w, kernel=8, grid=grid)
        close(out, wr.T, 1e-6, what=f"gemm 8pp (192-row tiles) identity grid {grid}")


def test_gemm_8pp_under_load_is_deterministic():
    """The counted waits only ever assume FEWER outstanding operations than there are; with a second stream hammering HBM (store
    acknowledgements and DMA arrivals come late and out of step) the results must still be the bits of an undisturbed launch."""
    dtype, M, N, K = "bf16", 6000, 2048, 1024
    a, _ = q(rnd(M, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    bias = rnd(N, seed=7).to(DEV)
    ref = gemm(dtype, a, w, bias=bias, act=1, kernel=5)
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for grid in (64, 128, 0):
        with torch.cuda.stream(side):
            for _ in range(40):
                junk.add_(1)
        for _ in range(6):
            out = gemm(dtype, a, w, bias=bias, act=1, kernel=3, grid=grid)
            assert torch.equal(out, ref), f"persistent GEMM under load (grid {grid}) differs from the 16-wave kernel"
        torch.cuda.synchronize()


def test_gemm_8pp_fused_swiglu_equals_gemm_plus_swiglu_kernel():
    dtype, M, I, K = "bf16", 2504, 4096, 256
    x, xr = q(rnd(M, K, seed=1), dtype)
    wg, _ = q(rnd(I, K, seed=2) / math.sqrt(K), dtype)
    wu, _ = q(rnd(I, K, seed=3) / math.sqrt(K), dtype)
    inter = torch.stack([wg.reshape(-1, 16, K), wu.reshape(-1, 16, K)], 1).reshape(2 * I, K).contiguous()
    gu = gemm(dtype, x, inter, kernel=5)
    two = torch.empty(M, I, dtype=tdt(dtype), device=DEV)
    L.check(L.load().oavsr_op_swiglu(dcode(dtype), P(gu), 2 * I, P(two), I, M, I, 1, stream()))
    for grid in (16, 0):
        fused = gemm(dtype, x, inter, swiglu_pairs=1, kernel=3, grid=grid)
        assert torch.equal(fused, two), f"persistent fused SwiGLU (grid {grid}) differs from gemm + swiglu: {(fused.float() - two.float()).abs().max().item():.3e}"


def test_gemm_dispatch_takes_persistent_kernel_for_multi_round_launches():
    """More 256 x 256 tiles than CUs, plain bf16 epilogue -> the dispatcher's choice is the persistent kernel; same bits as the 16-wave one."""
    dtype, M, N, K = "bf16", 12000, 4096, 1024
    a, _ = q(rnd(M, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    bias = rnd(N, seed=7).to(DEV)
    auto = gemm(dtype, a, w, bias=bias, act=1)
    forced = gemm(dtype, a, w, bias=bias, act=1, kernel=5)
    assert torch.equal(auto, forced)


@pytest.mark.parametrize("M,N,K,K2", [(2504, 2048, 2048, 0), (2504, 2048, 8192, 0), (2397, 2048, 1024, 64), (1500, 3072, 512, 0)])
def test_gemm_under_filled_launches_same_bits_on_every_tile(M, N, K, K2):
    """The LLM prefill's o_proj / down_proj / QKV (80 .. 120 tiles of 256 x 256: the dispatcher takes the coarse tile from 64 tiles on, for the
    job's CU-time): the dispatcher's choice and the forced 256 x 128 tile give the same bits -- residual, bias, a LoRA segment and a ragged
    last row tile included -- and match the fp32 reference."""
    dtype = "bf16"
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    r, rr = q(rnd(M, N, seed=5), dtype)
    bias = rnd(N, seed=7)
    kw, ref = {}, ar @ wr.T + bias + rr
    if K2:
        a2, a2r = q(rnd(M, K2, seed=6), dtype)
        w2, w2r = q(rnd(N, K2, seed=8) / math.sqrt(K2), dtype)
        kw = dict(A2=a2, W2=w2)
        ref = ref + a2r @ w2r.T
    auto = gemm(dtype, a, w, bias=bias.to(DEV), residual=r, **kw)
    t256 = gemm(dtype, a, w, bias=bias.to(DEV), residual=r, kernel=7, **kw)
    close(auto, ref, tol(dtype) * 2, what="under-filled launch")
    assert torch.equal(auto, t256)


@pytest.mark.parametrize("M,N,K,res", [(2504, 2048, 8192, True), (2504, 3072, 2048, False), (12000, 1024, 1024, True), (3200, 3072, 1088, False), (12000, 3072, 1024, False)])
def test_launch_objective_changes_tiles_not_bits(M, N, K, res):
    """oavsr_set_launch_objective: calls sharing the chip keep launches of 64 .. 255 coarse tiles on the persistent 256 x 256 kernel (least
    CU-time), one call at a time takes finer tiles / the 192-row form (soonest finish) -- the same bits, and the bits of the 256 x 128 tile."""
    lib = L.load()
    dtype = "bf16"
    a, ar = q(rnd(M, K, seed=1), dtype)
    w, wr = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    bias = rnd(N, seed=7)
    r = q(rnd(M, N, seed=5), dtype)[0] if res else None
    try:
        L.check(lib.oavsr_set_launch_objective(0))
        shared = gemm(dtype, a, w, bias=bias.to(DEV), residual=r)
        L.check(lib.oavsr_set_launch_objective(1))
        alone = gemm(dtype, a, w, bias=bias.to(DEV), residual=r)
    finally:
        L.check(lib.oavsr_set_launch_objective(0))
    t256 = gemm(dtype, a, w, bias=bias.to(DEV), residual=r, kernel=7)
    assert torch.equal(shared, alone) and torch.equal(shared, t256)


@pytest.mark.parametrize("dtype,M,N,K,K2", [("bf16", 3200, 3072, 1024, 64), ("bf16", 1000, 3072, 1024, 64), ("f32", 700, 384, 128, 32)])
def test_gemm_lora_segment_equals_one_concatenated_k_range(dtype, M, N, K, K2):
    """model.hip: run_encoder_layers hands the video encoder's QKV + LoRA GEMM over as ONE K range -- A = [h | t] (row stride K + K2),
    W = [W | s B] -- so that the persistent 256 x 256 kernel (no second K segment) can take it: whole 128-byte k-tiles on both sides, the same
    k-tile sequence, hence the same bits as the two-segment launch."""
    a, _ = q(rnd(M, K, seed=1), dtype)
    w, _ = q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    a2, _ = q(rnd(M, K2, seed=3), dtype)
    w2, _ = q(rnd(N, K2, seed=4) / math.sqrt(K2), dtype)
    bias = rnd(N, seed=7).to(DEV)
    two = gemm(dtype, a, w, A2=a2, W2=w2, bias=bias)
    one = gemm(dtype, torch.cat([a, a2], 1).contiguous(), torch.cat([w, w2], 1).contiguous(), bias=bias)
    assert torch.equal(one, two)


