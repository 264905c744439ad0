"""evp_gemm on the GPU: every layout / dtype / tile / epilogue against float64 matmul on the host.
Tolerances: f32 path 2e-5 relative to sum|a||b| (exact-f32 MFMA = fmaf chain); bf16 path exact on small-integer data
(products and sums representable) and 1e-2 relative on random data (bf16 inputs, f32 accumulate, bf16 output)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYOUTS = [(False, False), (False, True), (True, True)]


def _mk(M, N, K, ta, tb, dtype, gen, ints=False, lda_pad=0, ldb_pad=0):
    def rnd(*s):
        if ints:
            return torch.randint(-3, 4, s, generator=gen).float()
        return torch.randn(*s, generator=gen)
    a_log, b_log = rnd(M, K), rnd(N, K)
    a = (a_log.t() if ta else a_log).contiguous()
    b = (b_log.t() if tb else b_log).contiguous()
    if lda_pad:
        a = torch.nn.functional.pad(a, (0, lda_pad))
    if ldb_pad:
        b = torch.nn.functional.pad(b, (0, ldb_pad))
    a, b = a.to(dtype), b.to(dtype)
    a_log = (a[:, :a.shape[1] - lda_pad] if lda_pad else a).float()
    b_log = (b[:, :b.shape[1] - ldb_pad] if ldb_pad else b).float()
    a_log = a_log.t() if ta else a_log
    b_log = b_log.t() if tb else b_log
    return a, b, a_log.double(), b_log.double()


@pytest.mark.parametrize("ta,tb", LAYOUTS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tile", [1, 2])
def test_layouts_exact_on_integers(ta, tb, dtype, tile):
    """Asymmetric small-integer operands: any row/col or k-order mix-up in a fragment map shows as a wrong integer."""
    from eventpretrain_amd import ops
    gen = torch.Generator().manual_seed(3)
    for (M, N, K) in [(128, 128, 64), (200, 136, 96), (64, 64, 32), (16, 8, 8), (130, 72, 200), (257, 264, 72)]:
        lda_pad = (-K) % 8 if not ta else (-M) % 8
        ldb_pad = (-K) % 8 if not tb else (-N) % 8
        a, b, al, bl = _mk(M, N, K, ta, tb, dtype, gen, ints=True, lda_pad=lda_pad, ldb_pad=ldb_pad)
        out = torch.empty(M, N, dtype=torch.float32).cuda()
        ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=ta, trans_b=tb, lda=a.shape[1], ldb=b.shape[1], tile=tile)
        ref = (al @ bl.t()).float()
        assert torch.equal(out.cpu(), ref), (M, N, K, (out.cpu() - ref).abs().max())


@pytest.mark.parametrize("ta,tb", LAYOUTS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_random_and_ragged_shapes(ta, tb, dtype):
    from eventpretrain_amd import ops
    gen = torch.Generator().manual_seed(4)
    shapes = [(6272 // 8, 768, 768), (98, 98, 64), (196, 32, 196), (1, 8, 8), (333, 24, 40), (100, 1000, 16), (392, 4096, 384)]
    for (M, N, K) in shapes:
        # contiguous dims must be multiples of 8 (bf16) / 4 (f32): pad the operand whose ragged dim is contiguous
        lda_pad = (-K) % 8 if not ta else (-M) % 8
        ldb_pad = (-K) % 8 if not tb else (-N) % 8
        a, b, al, bl = _mk(M, N, K, ta, tb, dtype, gen, lda_pad=lda_pad, ldb_pad=ldb_pad)
        ldc = (N + 7) // 8 * 8
        for out_dtype in (torch.float32, dtype):
            out = torch.full((M, ldc), 7.0, dtype=out_dtype).cuda()
            ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=ta, trans_b=tb, lda=a.shape[1], ldb=b.shape[1], ldc=ldc)
            ref = al @ bl.t()
            scale = (al.abs() @ bl.abs().t()).clamp_min(1e-6)
            err = ((out.cpu().double()[:, :N] - ref).abs() / scale).max().item()
            tol = 2e-5 if (dtype == torch.float32 and out_dtype == torch.float32) else 1e-2
            assert err <= tol, (M, N, K, out_dtype, err)
            if ldc > N:
                assert (out.cpu()[:, N:] == 7.0).all(), "wrote outside N"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_epilogues(dtype):
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_RELU
    gen = torch.Generator().manual_seed(5)
    M, N, K = 200, 136, 96
    a, b, al, bl = _mk(M, N, K, False, False, dtype, gen)
    bias = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen)
    base = 0.5 * (al @ bl.t()) + bias.double()
    tol = 1e-4 if dtype == torch.float32 else 2e-2

    def close(x, y, t=tol):
        return ((x.cpu().double() - y).abs() / (y.abs() + 1.0)).max().item() <= t

    # bias + GELU with pre-activation aux + residual
    out = torch.empty(M, N, dtype=torch.float32).cuda()
    aux = torch.empty(M, N, dtype=torch.float32).cuda()
    ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), act=ACT_GELU, aux=aux, residual=res.cuda())
    assert close(aux, base)
    assert close(out, torch.nn.functional.gelu(base) + res.double())
    # ReLU
    ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), act=ACT_RELU)
    assert close(out, torch.relu(base))
    # dGELU / dReLU read aux
    h = torch.randn(M, N, generator=gen)
    hd = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hd).sum().backward()
    bt = b.t().contiguous()          # activation backward is built for the dgrad layout (transB)
    ops.gemm(a.cuda(), bt.cuda(), out, M=M, N=N, K=K, trans_b=True, act=ACT_DGELU, aux=h.cuda())
    assert close(out, (al @ bl.t()) * hd.grad)
    ops.gemm(a.cuda(), bt.cuda(), out, M=M, N=N, K=K, trans_b=True, act=ACT_DRELU, aux=h.cuda())
    assert close(out, (al @ bl.t()) * (h > 0).double())
    if dtype == torch.bfloat16:
        hb, ob = h.to(dtype), torch.empty(M, N, dtype=dtype).cuda()
        ops.gemm(a.cuda(), bt.cuda(), ob, M=M, N=N, K=K, trans_b=True, act=ACT_DGELU, aux=hb.cuda())
        hd2 = hb.float().double().requires_grad_(True)
        torch.nn.functional.gelu(hd2).sum().backward()
        assert close(ob.float(), (al @ bl.t()) * hd2.grad, 2e-2)
    # accumulate
    acc0 = torch.randn(M, N, generator=gen)
    out = acc0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, accumulate=True)
    assert close(out, acc0.double() + al @ bl.t())
    # low-precision output with aux
    if dtype == torch.bfloat16:
        out = torch.empty(M, N, dtype=dtype).cuda()
        aux = torch.empty(M, N, dtype=dtype).cuda()
        ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), act=ACT_GELU, aux=aux)
        assert close(aux.float(), base, 2e-2) and close(out.float(), torch.nn.functional.gelu(base), 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_strided(dtype):
    """The attention-core shapes: operands are slices of a packed [B,N,3,h,dh] tensor."""
    from eventpretrain_amd import ops
    gen = torch.Generator().manual_seed(6)
    B, Nn, h, dh = 3, 50, 4, 32
    Cc = h * dh
    qkv = torch.randn(B, Nn, 3, h, dh, generator=gen).to(dtype)
    q, k, v = [qkv[:, :, i].float().double().permute(0, 2, 1, 3) for i in range(3)]       # (B,h,N,dh)
    ldp = (Nn + 7) // 8 * 8
    s = torch.zeros(B, h, Nn, ldp, dtype=torch.float32).cuda()
    qd = qkv.cuda()
    ops.gemm(qd, qd, s, M=Nn, N=Nn, K=dh, lda=3 * Cc, ldb=3 * Cc, ldc=ldp, b_off=Cc, alpha=0.25, batch=(B, h),
             stride_a=(Nn * 3 * Cc, dh), stride_b=(Nn * 3 * Cc, dh), stride_c=(h * Nn * ldp, Nn * ldp))
    ref = 0.25 * q @ k.transpose(-1, -2)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert ((s.cpu().double()[..., :Nn] - ref).abs() / (q.abs() @ k.abs().transpose(-1, -2) + 1e-3)).max() <= tol
    # out = p v with v k-strided (transB) and the output scattered back to [B,N,h*dh]
    p = torch.rand(B, h, Nn, ldp, generator=gen)
    p[..., Nn:] = 0
    p = p.to(dtype)
    o = torch.zeros(B, Nn, Cc, dtype=torch.float32).cuda()
    ops.gemm(p.cuda(), qd, o, M=Nn, N=dh, K=Nn, lda=ldp, ldb=3 * Cc, ldc=Cc, b_off=2 * Cc, trans_b=True, batch=(B, h),
             stride_a=(h * Nn * ldp, Nn * ldp), stride_b=(Nn * 3 * Cc, dh), stride_c=(Nn * Cc, dh))
    refo = (p.float().double()[..., :Nn] @ v).permute(0, 2, 1, 3).reshape(B, Nn, Cc)
    assert ((o.cpu().double() - refo).abs() / (refo.abs() + 1.0)).max() <= tol


def test_split_k_matches_single_pass():
    """Weight-gradient shape (few output tiles, long K): split-K with LDS-staged f32 atomics vs one pass."""
    from eventpretrain_amd import ops
    gen = torch.Generator().manual_seed(8)
    for dtype in (torch.bfloat16, torch.float32):
        M, N, K = 384, 256, 4096 + 40
        a, b, al, bl = _mk(M, N, K, True, True, dtype, gen)
        ref = al @ bl.t()
        for sk in (0, 1, 3, 7):
            out = torch.full((M, N), 5.0).cuda()
            ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, splitk=sk, tile=1)
            scale = (al.abs() @ bl.abs().t())
            assert ((out.cpu().double() - ref).abs() / scale).max() <= (2e-5 if dtype == torch.float32 else 1e-2), sk
        acc0 = torch.randn(M, N, generator=gen)
        out = acc0.clone().cuda()
        ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, splitk=4, accumulate=True, tile=1)
        assert ((out.cpu().double() - ref - acc0.double()).abs() / (al.abs() @ bl.abs().t())).max() <= 1e-2


def test_argument_errors_raise():
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import EvpError
    a = torch.zeros(8, 12).cuda()
    with pytest.raises(EvpError):
        ops.gemm(a, a, torch.zeros(8, 8).cuda(), M=8, N=8, K=12, lda=10)          # lda not a multiple of 4
    with pytest.raises(EvpError):
        ops.gemm(a, a, torch.zeros(8, 8).cuda(), M=8, N=8, K=12, trans_a=True)     # (1,0) layout not built
    with pytest.raises(EvpError):
        ops.gemm(a.cpu(), a.cpu(), torch.zeros(8, 8), M=8, N=8, K=12)              # no CPU path


@pytest.mark.parametrize("entry", ["evp_gemm_grouped_tn_g4_bf16"])
def test_grouped_tn256_with_fused_column_sums_exact(entry):
    """evp_gemm_grouped_tn_g4_bf16 (the G4 one-wave-per-SIMD body): several dW = dY^T X problems in one launch of a
    256x256-tile kernel, with the bias
    gradient (column sums of dY) produced by the same kernel; small-integer data, so every result is exact. Covers ragged
    M / N (not multiples of 256), accumulate into an existing dW / db, and a problem without column sums."""
    import numpy as np
    from eventpretrain_amd._lib import call, stream_ptr
    g = torch.Generator(device="cuda").manual_seed(11)
    specs = [(768, 512, 1280, True, False), (304, 264, 192, True, True), (256, 256, 128 if "g4" in entry else 64, False, False),
             (2304, 768, 6272, True, False), (520, 776, 96 if "g4" in entry else 128, True, True)]
    pdt = np.dtype([("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("lda", "<i4"),
                    ("ldb", "<i4"), ("ldc", "<i4"), ("acc", "<i4"), ("cacc", "<i4"), ("colsum", "<u8")])
    probs = np.zeros(len(specs), dtype=pdt)
    keep, items = [], []
    for i, (M, N, K, want_cs, acc) in enumerate(specs):
        dy = torch.randint(-2, 3, (K, M), generator=g, device="cuda").bfloat16()
        x = torch.randint(-2, 3, (K, N), generator=g, device="cuda").bfloat16()
        dw = torch.full((M, N), 3.0, device="cuda")
        db = torch.full((M,), 5.0, device="cuda")
        keep.append((dy, x, dw, db, want_cs, acc))
        probs[i] = (dy.data_ptr(), x.data_ptr(), dw.data_ptr(), M, N, K, M, N, N, int(acc), int(acc), db.data_ptr() if want_cs else 0)
        for tn in range((N + 255) // 256):
            for tm in range((M + 255) // 256):
                items.append((i, tm, tn, 0))
    items.insert(3, (-1, 0, 0, 0))          # padding items (per-XCD list layout) are skipped
    items.append((-1, 0, 0, 0))
    pt = torch.from_numpy(probs.view(np.uint8)).cuda()
    it = torch.tensor(items, dtype=torch.int32, device="cuda")
    call(entry, pt.data_ptr(), it.data_ptr(), len(items), stream_ptr())
    torch.cuda.synchronize()
    for dy, x, dw, db, want_cs, acc in keep:
        ref_w = dy.float().t() @ x.float() + (3.0 if acc else 0.0)
        assert torch.equal(dw, ref_w)
        if want_cs:
            assert torch.equal(db, dy.float().sum(0) + (5.0 if acc else 0.0))
        else:
            assert torch.equal(db, torch.full_like(db, 5.0))


def test_g4_tn_tile_exact_and_random():
    """evp_gemm tile 9: the G4 body as a plain TN GEMM -- exact on small integers (ragged M / N, K = 96 .. 6272, accumulate),
    and within f32-accumulation tolerance of float64 on random data."""
    from eventpretrain_amd import ops
    g = torch.Generator().manual_seed(3)
    for M, N, K in [(256, 256, 96), (304, 520, 160), (768, 3072, 6272), (1000, 264, 1056)]:
        a = torch.randint(-3, 4, (K, M), generator=g).to(torch.bfloat16)
        b = torch.randint(-3, 4, (K, N), generator=g).to(torch.bfloat16)
        ref = a.double().t() @ b.double()
        out = torch.full((M, N), 2.0, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, tile=9)
        assert torch.equal(out.cpu().double(), ref), (M, N, K)
        ops.gemm(a.cuda(), b.cuda(), out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, tile=9, accumulate=True)
        assert torch.equal(out.cpu().double(), 2 * ref), (M, N, K, "accumulate")
    a = torch.randn(2048, 520, generator=g).to(torch.bfloat16)
    b = torch.randn(2048, 392, generator=g).to(torch.bfloat16)
    out = torch.empty(520, 392, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), out, M=520, N=392, K=2048, trans_a=True, trans_b=True, lda=520, ldb=392, tile=9)
    ref = a.double().t() @ b.double()
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() * 45
    with pytest.raises(Exception):
        ops.gemm(a.cuda(), b.cuda(), out, M=520, N=392, K=2048, lda=2048, ldb=2048, tile=9)      # not the TN layout


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("tile", [20, 21, 22])
def test_g4_forward_dgrad_tiles_exact_on_integers(tb, tile):
    """evp_gemm tiles 20 / 21 / 22 (G4 bodies for forward = NT and data gradient = NN: 256x256, 256x128, 128x256; bf16 C parked
    in LDS and written as whole rows): exact on small-integer operands, ragged M (not a multiple of the tile) and ragged N
    (multiple of 8 only), K from the minimum (96) up, untouched columns beyond N, and a run-to-run identical result."""
    from eventpretrain_amd import ops
    gen = torch.Generator().manual_seed(41 + tile)
    for (M, N, K) in [(256, 256, 96), (520, 264, 160), (1000, 1032, 768), (6272, 2304, 768), (300, 8, 128)]:
        a, b, al, bl = _mk(M, N, K, False, tb, torch.bfloat16, gen, ints=True)
        ldc = N + 8
        out = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device="cuda")
        kw = dict(M=M, N=N, K=K, trans_b=tb, ldb=(N if tb else K), ldc=ldc, tile=tile)
        ops.gemm(a.cuda(), b.cuda(), out, **kw)
        ref = (al @ bl.t()).float().to(torch.bfloat16).double()      # integer sums are exact in the f32 accumulator: one rounding, to bf16
        assert torch.equal(out[:, :N].cpu().double(), ref), (M, N, K, (out[:, :N].cpu().double() - ref).abs().max())
        assert (out[:, N:].float() == 7.0).all(), "wrote outside N"
        out2 = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), out2, **kw)
        assert torch.equal(out, out2)


@pytest.mark.parametrize("tile", [20, 21, 22])
def test_g4_epilogues_match_the_128_tile_kernel(tile):
    """Bias, GELU + pre-activation store (forward layout) and GELU' (data-gradient layout) of the G4 bodies on random data:
    against float64, and against the 128x128 kernel (same operand rounding, same polynomial GELU: equal up to the f32
    summation order inside the MFMA chain, i.e. at most one bf16 ulp of the output)."""
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU, ACT_RELU, ACT_DRELU
    gen = torch.Generator().manual_seed(50 + tile)
    M, N, K = 1544, 1032, 416

    def ulp_close(x, y):                       # bf16 outputs: within 2 ulp of each other
        x, y = x.float(), y.float()
        return bool(((x - y).abs() <= 2.0 ** -7 * torch.maximum(x.abs(), y.abs()) + 1e-6).all())

    a, b, al, bl = _mk(M, N, K, False, False, torch.bfloat16, gen)
    b = (b * 0.1)
    bl = b.float().double()
    bias = torch.randn(N, generator=gen)
    base = 0.5 * (al @ bl.t()) + bias.double()
    outs = {}
    for t in (1, tile):
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), c, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), tile=t)
        h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        x = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), h, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), act=ACT_GELU, aux=x, tile=t)
        r = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), r, M=M, N=N, K=K, alpha=0.5, bias=bias.cuda(), act=ACT_RELU, tile=t)
        outs[t] = (c, h, x, r)
    for got, ref in zip(outs[tile], (base, torch.nn.functional.gelu(base), base, torch.relu(base))):
        assert ((got.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max().item() <= 2e-2
    for g_, o_ in zip(outs[tile], outs[1]):
        assert ulp_close(g_, o_)
    # data-gradient layout with GELU' / ReLU' of a stored pre-activation
    bt = b.t().contiguous()
    hpre = torch.randn(M, N, generator=gen).to(torch.bfloat16)
    hd = hpre.float().double().requires_grad_(True)
    torch.nn.functional.gelu(hd).sum().backward()
    res = {}
    for t in (1, tile):
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), bt.cuda(), o, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=hpre.cuda(), tile=t)
        o2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a.cuda(), bt.cuda(), o2, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DRELU, aux=hpre.cuda(), tile=t)
        res[t] = (o, o2)
    ref = (al @ bl.t()) * hd.grad
    assert ((res[tile][0].cpu().double() - ref).abs() / (ref.abs() + 1.0)).max().item() <= 2e-2
    assert ulp_close(res[tile][0], res[1][0]) and ulp_close(res[tile][1], res[1][1])


def test_g4_forward_tiles_refuse_what_they_do_not_take():
    from eventpretrain_amd import EvpError, ops
    a = torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(EvpError):          # f32 C
        ops.gemm(a, a, torch.empty(256, 256, device="cuda"), M=256, N=256, K=128, tile=20)
    with pytest.raises(EvpError):          # K not a multiple of 32
        ops.gemm(a, a, torch.empty(256, 256, dtype=torch.bfloat16, device="cuda"), M=256, N=256, K=72, lda=128, ldb=128, tile=21)
    with pytest.raises(EvpError):          # removed variants say so
        ops.gemm(a, a, torch.empty(256, 256, device="cuda"), M=256, N=256, K=128, tile=12)


def test_wide_outputs_take_the_g4_bodies_when_switched_on():
    """evp_gemm_set_variant(11): the automatic tile choice sends one-round wide forward GEMMs (256x256 tiles) and wide data gradients
    (128x256) to the G4 bodies -- same result as forcing tile 20 / 22; variant 10 (the default: the G4 bodies lose inside the
    replayed step) keeps them on 128x128 tiles."""
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import call
    gen = torch.Generator().manual_seed(77)
    for (M, N, K, tb, forced) in [(2304, 1024, 256, False, 20), (12544, 1536, 128, True, 22)]:
        a, b, _, _ = _mk(M, N, K, False, tb, torch.bfloat16, gen)
        c0, c1, c2 = (torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        kw = dict(M=M, N=N, K=K, trans_b=tb, ldb=(N if tb else K))
        call("evp_gemm_set_variant", 11)
        try:
            ops.gemm(a.cuda(), b.cuda(), c0, **kw)
        finally:
            call("evp_gemm_set_variant", 10)
        ops.gemm(a.cuda(), b.cuda(), c1, tile=forced, **kw)
        assert torch.equal(c0, c1)
        ops.gemm(a.cuda(), b.cuda(), c2, **kw)
        ops.gemm(a.cuda(), b.cuda(), c1, tile=1, **kw)
        assert torch.equal(c1, c2)
        assert (c0.float() - c2.float()).abs().max().item() <= 2.0 ** -7 * c0.float().abs().max().item()


def test_eight_column_write_through_epilogue_is_bit_identical_and_falls_back():
    """bf16 tiles that lie wholly inside C leave the epilogue in 8-column pieces as 16-byte write-through stores
    (gemm.hip epilogue_lds_rows8; evp_gemm_set_variant(18) keeps the 4-column form). Same arithmetic per element: bit-identical for
    bias, GELU + stored pre-activation, ReLU, GELU' -- on a shape with whole and ragged tiles (the ragged ones take the old path
    inside the same launch), on the 96x128 tiling, and with a row stride / base pointer the 16-byte form must refuse (ldc % 8 != 0,
    aux 8 bytes off): then the launcher falls back by itself and the result is still right."""
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU, ACT_RELU, call
    gen = torch.Generator().manual_seed(123)
    for (M, N, K) in [(1000, 600, 160), (6272, 768, 96), (640, 512, 64)]:
        a, b, al, bl = _mk(M, N, K, False, False, torch.bfloat16, gen)
        bt = b.t().contiguous()
        bias = torch.randn(N, generator=gen).cuda()
        hpre = torch.randn(M, N, generator=gen).to(torch.bfloat16).cuda()
        a, b, bt = a.cuda(), b.cuda(), bt.cuda()

        def run():
            c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            ops.gemm(a, b, c, M=M, N=N, K=K, bias=bias)
            h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            x = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            ops.gemm(a, b, h, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=x)
            r = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            ops.gemm(a, b, r, M=M, N=N, K=K, bias=bias, act=ACT_RELU)
            g = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            ops.gemm(a, bt, g, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=hpre)
            return c, h, x, r, g

        new = run()
        call("evp_gemm_set_variant", 18)
        try:
            old = run()
        finally:
            call("evp_gemm_set_variant", 19)
        for t0, t1 in zip(old, new):
            assert torch.equal(t0, t1)
        ref = torch.nn.functional.gelu(al @ bl.t() + bias.cpu().double())
        assert ((new[1].cpu().double() - ref).abs() / (ref.abs() + 1.0)).max().item() <= 2e-2
        # strides / bases the 16-byte form cannot take
        wide = torch.zeros(M, N + 4, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a, b, wide, M=M, N=N, K=K, bias=bias, ldc=N + 4)
        assert torch.equal(wide[:, :N], new[0]) and (wide[:, N:] == 0).all()
        buf = torch.zeros(M * N + 4, dtype=torch.bfloat16, device="cuda")
        aux_off = buf[4:].view(M, N)                      # 8 bytes off a 16-byte boundary
        h2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(a, b, h2, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux_off)
        assert torch.equal(h2, new[1]) and torch.equal(aux_off, new[2])
