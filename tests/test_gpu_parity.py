"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every call goes through the
C-ABI library (ctypes); the oracle / golden fixtures are the checker only."""
import numpy as np
import pytest
import torch

from tests.conftest import golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

LOGIT_TOL = 1e-3   # BASELINE.json: logits within 1e-3 fp32 of the reference CPU path
MEL_TOL = 1e-3     # log-mel domain (SURVEY.md section 7 minimum slice)


def dev():
    return torch.device("cuda:0")


def _load(model, sd):
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own, k
        own[k] = torch.from_numpy(np.array(v, copy=True))
    model.load_state_dict(own)
    return model.to(dev())


@pytest.fixture(scope="module")
def sd_model(sd_weights):
    from tal_asrd_amd import SDModel
    return _load(SDModel(), sd_weights)


@pytest.fixture(scope="module")
def asr_model(asr_weights):
    from tal_asrd_amd import ASRModel
    return _load(ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True), asr_weights)


# ------------------------------------------------------------------ dense layer
@pytest.mark.parametrize("M,N,K,mode", [
    (128, 160, 32, 0), (1, 1, 4, 0), (130, 161, 36, 1), (257, 800, 800, 2), (64, 6008, 128, 0),
    (513, 1440, 1440, 1), (7, 512, 2048, 2), (300, 24, 16, 0),
])
def test_linear_matches_fp64(M, N, K, mode):
    """Asymmetric random operands (transpose-detecting), fp64 host reference."""
    from tal_asrd_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    if mode == 1:
        ref = ref.clamp_min(0)
    if mode == 2:
        ref = res.double() + 0.3 * ref
    y = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), mode=mode, res=res.to(dev()) if mode == 2 else None,
                   alpha=0.3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), atol=2e-5 * max(1.0, K ** 0.5 / 8), rtol=1e-5)


@pytest.mark.parametrize("M,N,K,mode,split", [
    (70000, 320, 256, 1, True),      # 1094 tiles = 2 whole rounds + 70 tail tiles (cut along K)
    (100000, 160, 512, 2, False),    # 782 tiles: 270 tail tiles; the cost model declines (scratch traffic > gain)
    (100000, 160, 1440, 2, True),    # same tiles, long K: 270 tail tiles in 3 slices each
    (66000, 161, 288, 0, True),      # ragged N, K/32 = 9 K steps
    (65536 + 128, 160, 256, 3, True),  # a single tail tile
])
def test_linear_split_k_tail_matches_fp64_and_is_deterministic(M, N, K, mode, split):
    """The scratch-enabled dense layer (tal_linear_ws_fwd): tiles of the last partial round are summed from K
    slices by the fix-up kernel -- same tolerance as the plain path, bitwise repeatable, and identical to the
    plain path on every row that is not in a tail tile."""
    import ctypes as C
    from tal_asrd_amd import ops, _native as N_
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    res = torch.randn(M, N, generator=g).to(dev())
    lib = N_.lib()
    assert lib.tal_linear_workspace_bytes(M, N, K) > 0
    y1 = ops.linear(x, w, b, mode=mode, res=res if mode == 2 else None, alpha=0.3)
    y2 = ops.linear(x, w, b, mode=mode, res=res if mode == 2 else None, alpha=0.3)
    assert torch.equal(y1, y2)
    plain = torch.empty_like(y1)
    N_.check(lib.tal_linear_fwd(N_.ptr(x), N_.ptr(w), N_.ptr(b), N_.ptr(res if mode == 2 else None), 0.3, mode, M, N, K,
                                N_.ptr(plain), N_.stream_handle()), "tal_linear_fwd")
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t() + b.double()
    if mode == 1:
        ref = ref.clamp_min(0)
    if mode == 2:
        ref = res.double() + 0.3 * ref
    if mode == 3:
        ref = 0.3 * ref
    tol = 2e-5 * max(1.0, K ** 0.5 / 8)
    assert float((y1.double() - ref).abs().max()) < tol
    assert float((plain.double() - ref).abs().max()) < tol
    same_rows = (y1 == plain).all(dim=1)
    assert int(same_rows.sum()) >= 65536 - 128          # the whole-round tiles are untouched by the split
    assert bool(same_rows.all()) != split               # ...and the tail went through the K slices iff expected


@pytest.mark.parametrize("M,N,K,mode", [(513 + 23, 160, 64, 0), (66000 + 55, 320, 256, 2), (1000 + 17, 800, 800, 1),
                                        (640 + 9, 1440, 1440, 2), (44983, 160, 256, 3)])
def test_linear_never_writes_past_the_last_row(M, N, K, mode):
    """Ragged M: the epilogue's buffer descriptors end at row M-1; the rows behind y (and behind the residual)
    must stay untouched, for whole tiles, tail tiles and the split-K fix-up alike."""
    from tal_asrd_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    guard = 7
    yfull = torch.full((M + guard, N), 12345.0, device=dev())
    resfull = torch.full((M + guard, N), float("nan"), device=dev())
    resfull[:M] = torch.randn(M, N, generator=g).to(dev())
    y = ops.linear(x, w, b, mode=mode, res=resfull[:M] if mode == 2 else None, alpha=0.3, out=yfull[:M])
    torch.cuda.synchronize()
    assert bool((yfull[M:] == 12345.0).all())
    assert bool(torch.isfinite(y).all())
    ref = x.double() @ w.double().t() + b.double()
    if mode == 1:
        ref = ref.clamp_min(0)
    if mode == 2:
        ref = resfull[:M].double() + 0.3 * ref
    if mode == 3:
        ref = 0.3 * ref
    assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, K ** 0.5 / 8)


def _np_split(x):
    """host statement of the fp16x3 operand format (include/tal_asrd.h): per row and 32-wide K block, 32 hi halves
    then 32 lo halves, hi = fp16(x), lo = fp16((x - hi) * 2^11)."""
    rows, K = x.shape
    hi = np.clip(x, -65504.0, 65504.0).astype(np.float16)
    lo = np.clip((x - hi.astype(np.float32)) * np.float32(2048.0), -65504.0, 65504.0).astype(np.float16)
    out = np.empty((rows, K // 32, 64), dtype=np.float16)
    out[:, :, :32] = hi.reshape(rows, K // 32, 32)
    out[:, :, 32:] = lo.reshape(rows, K // 32, 32)
    return out


def test_split_f16x3_format_is_bit_exact():
    from tal_asrd_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(777, 160, generator=g) * torch.logspace(-6, 6, 160)).contiguous()   # incl. values past the fp16 range
    got = ops.split_f16x3(x.to(dev())).cpu().numpy().view(np.float16).reshape(777, 5, 64)
    np.testing.assert_array_equal(got.view(np.uint16), _np_split(x.numpy()).view(np.uint16))


@pytest.mark.parametrize("M,C,K", [(70000, 320, 256), (66000 + 55, 160, 1440), (1000 + 17, 800, 800), (44983, 1440, 1440)])
def test_linear_f16x3_pair_matches_fp64(M, C, K):
    """The two dense layers of a TDSBlock in the fp16x3 form (relu layer writing its output pre-split, residual layer
    reading it): against float64, with the tolerance of the fp32 path, guard rows untouched, bitwise repeatable."""
    import ctypes as CT
    from tal_asrd_amd import ops, _native as N_
    lib = N_.lib()
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, K, generator=g).to(dev())
    w0 = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev())
    b0 = torch.randn(C, generator=g).to(dev())
    w1 = (torch.randn(K, C, generator=g) / C ** 0.5).to(dev())
    b1 = torch.randn(K, generator=g).to(dev())
    xs, w0s, w1s = ops.split_f16x3(x), ops.split_f16x3(w0), ops.split_f16x3(w1)
    nws = lib.tal_linear_workspace_bytes(M, C, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev())
    guard = 5
    hs = torch.full(((M + guard) * C * 4,), 0x5A, dtype=torch.uint8, device=dev())
    yfull = torch.full((M + guard, K), 4321.0, device=dev())
    def run():
        N_.check(lib.tal_linear_f16x3_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), None, 0.0, 1, M, C, K, N_.ptr(hs), 1, N_.ptr(ws), nws,
                                          N_.stream_handle()), "tal_linear_f16x3_fwd")
        N_.check(lib.tal_linear_f16x3_fwd(N_.ptr(hs), N_.ptr(w1s), N_.ptr(b1), N_.ptr(x), 0.3, 2, M, K, C, N_.ptr(yfull), 0, N_.ptr(ws), nws,
                                          N_.stream_handle()), "tal_linear_f16x3_fwd")
        torch.cuda.synchronize()
        return yfull[:M].clone()
    y1 = run()
    y2 = run()
    assert torch.equal(y1, y2)
    assert bool((hs[M * C * 4:] == 0x5A).all()) and bool((yfull[M:] == 4321.0).all())
    h = torch.relu(x.double() @ w0.double().t() + b0.double())
    ref = x.double() + 0.3 * (h @ w1.double().t() + b1.double())
    err = float((y1.double() - ref).abs().max())
    assert err < 2e-5 * max(1.0, max(K, C) ** 0.5 / 8), err
    # the hidden activations, decoded from the split form
    hsp = hs[:M * C * 4].cpu().numpy().view(np.float16).reshape(M, C // 32, 64).astype(np.float32)
    hdec = (hsp[:, :, :32] + hsp[:, :, 32:] / 2048.0).reshape(M, C)
    assert float(np.abs(hdec - h.cpu().numpy()).max()) < 2e-5 * max(1.0, K ** 0.5 / 8)


@pytest.mark.parametrize("M,C", [(256 * 40 + 3, 1440), (256 * 71, 800)])
def test_linear_f16x3_256_row_kernel_equals_128_row_kernel(M, C):
    """The one-wave-per-SIMD kernel (csrc/gemm_w64.hip, 256 x 160 tiles) keeps the 128 x 160 kernel's arithmetic order per
    accumulator: with the K-sliced tail switched off the two produce BIT-IDENTICAL outputs -- fp32 and split form, relu layer
    and split-residual layer under the range guard -- and with their (differently cut) tails they agree to fp32 rounding."""
    import ctypes as CT
    from tal_asrd_amd import ops, _native as N_
    lib = N_.lib()
    g = torch.Generator().manual_seed(M + C)
    K = C
    x = torch.randn(M, K, generator=g).to(dev())
    w0 = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev())
    b0 = torch.randn(C, generator=g).to(dev())
    xs, w0s = ops.split_f16x3(x), ops.split_f16x3(w0)
    nws = lib.tal_linear_workspace_bytes(M, C, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev())
    flag = torch.zeros(16, dtype=torch.int32, device=dev())

    def run(mode, out_split, guarded):
        y = torch.zeros(M * C * 4, dtype=torch.uint8, device=dev())
        if guarded:
            N_.check(lib.tal_linear_f16x3_guarded_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), N_.ptr(xs) if mode == 2 else None, 1 if mode == 2 else 0,
                                                      0.3, mode, M, C, K, N_.ptr(y), out_split, N_.ptr(flag), N_.ptr(ws), nws, N_.stream_handle()),
                     "tal_linear_f16x3_guarded_fwd")
        else:
            N_.check(lib.tal_linear_f16x3_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), N_.ptr(x) if mode == 2 else None, 0.3, mode, M, C, K, N_.ptr(y),
                                              out_split, N_.ptr(ws), nws, N_.stream_handle()), "tal_linear_f16x3_fwd")
        torch.cuda.synchronize()
        return y

    def decode(y, out_split):
        if not out_split:
            return y.view(torch.float32).reshape(M, C)
        h = y.view(torch.float16).reshape(M, C // 32, 64).float()
        return (h[:, :, :32] + h[:, :, 32:] / 2048.0).reshape(M, C)

    cases = [(1, 0, False), (2, 0, False), (1, 1, True), (2, 1, True), (2, 0, True)]
    try:
        N_.set_option("gemm_no_splitk_tail", 1)
        for mode, out_split, guarded in cases:
            N_.set_option("gemm_no_w64", 0)
            a = run(mode, out_split, guarded)
            N_.set_option("gemm_no_w64", 1)
            b = run(mode, out_split, guarded)
            assert torch.equal(a, b), (mode, out_split, guarded)
        N_.set_option("gemm_no_splitk_tail", 0)
        for mode, out_split, guarded in cases:
            N_.set_option("gemm_no_w64", 0)
            a = decode(run(mode, out_split, guarded), out_split)
            N_.set_option("gemm_no_w64", 1)
            b = decode(run(mode, out_split, guarded), out_split)
            assert float((a - b).abs().max()) < 2e-5, (mode, out_split, guarded)
    finally:
        N_.set_option("gemm_no_w64", 0)
        N_.set_option("gemm_no_splitk_tail", 0)
    assert int(flag[0]) == 0


def test_linear_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed C write."""
    from tal_asrd_amd import ops
    K = 160
    w = torch.arange(K * K, dtype=torch.float32).reshape(K, K) / 1000.0
    y = ops.linear(torch.eye(K).to(dev()), w.to(dev()))
    np.testing.assert_array_equal(y.cpu().numpy(), w.t().numpy())


# ------------------------------------------------------------------ log-mel
@pytest.mark.parametrize("B,L", [(1, 16000), (2, 15999), (1, 480000), (3, 4000), (1, 201), (1, 5281)])
def test_logmel_matches_oracle(B, L):
    from oracle import tal_oracle as O
    from tal_asrd_amd import LogMelSpec, synth
    audio = synth.synth_audio_batch(B, L, 99)
    m = LogMelSpec().to(dev())
    y = m(torch.from_numpy(audio).to(dev()))
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, 1 + L // 160, 80)
    ref32 = O.logmel(audio).numpy()
    ref64 = O.logmel_f64(audio)
    got = y.cpu().numpy()
    assert np.abs(got - ref32).max() < MEL_TOL
    assert np.abs(got - ref64).max() < MEL_TOL
    assert abs(float(got.astype(np.float64).mean())) < 1e-5   # global mean removed


def test_logmel_silence_and_stats():
    """All-zero audio -> every value log(eps), mean subtracted -> exactly 0; stats expose (sum,count)."""
    from tal_asrd_amd import LogMelSpec, ops
    m = LogMelSpec().to(dev())
    a = torch.zeros(2, 8000, device=dev())
    y = m(a)
    assert float(y.abs().max()) < 1e-6
    raw, mean, stats = ops.logmel(m.plan(), a, subtract_mean=False, return_stats=True)
    np.testing.assert_allclose(raw.cpu().numpy(), np.log(np.float32(1e-6)), rtol=1e-6)
    assert stats[1].item() == 2 * 51 * 80
    np.testing.assert_allclose(stats[0].item() / stats[1].item(), mean.item(), rtol=1e-6)


def _logmel_f64(audio, window, fb, eps):
    """float64 restatement for an ARBITRARY window / filterbank (reflect pad 200, frames of 400 @ 160, rfft, power, fb, log)."""
    a = np.asarray(audio, dtype=np.float64)
    ap = np.pad(a, ((0, 0), (200, 200)), mode="reflect")
    T = 1 + a.shape[1] // 160
    idx = np.arange(400)[None, :] + 160 * np.arange(T)[:, None]
    spec = np.fft.rfft(ap[:, idx] * np.asarray(window, dtype=np.float64)[None, None, :], axis=-1)
    return np.log((spec.real ** 2 + spec.imag ** 2) @ np.asarray(fb, dtype=np.float64) + eps)


@pytest.mark.parametrize("B,L", [(1, 16000), (3, 48000), (2, 5281), (1, 201)])
def test_logmel_both_forms_match_the_oracle(B, L):
    """The front-end has two forms: the fast transform on the float64 vector ALU (default: two frames per complex 400 = 20 x 20
    transform, csrc/logmel.hip logmel_fft_kernel) and the float64 matrix-core DFT of rounds 1-4 (tal_set_option("logmel_mfma", 1)).
    Both against the float32 and float64 oracles; the fast form within 5e-5 of the float64 one (it applies the float32 window as given,
    the matrix form symmetrises it); a half-precision waveform gives exactly what the same samples give as float32."""
    from oracle import tal_oracle as O
    from tal_asrd_amd import LogMelSpec, synth, ops, _native as N_
    audio = synth.synth_audio_batch(B, L, 321)
    m = LogMelSpec().to(dev())
    x = torch.from_numpy(audio).to(dev())
    ref32, ref64 = O.logmel(audio, subtract_mean=False).numpy(), O.logmel_f64(audio, subtract_mean=False)
    try:
        for form in (0, 1):
            N_.set_option("logmel_mfma", form)
            got, mean, stats = ops.logmel(m.plan(), x, eps=m.eps, subtract_mean=False, return_stats=True)
            g = got.cpu().numpy()
            assert np.abs(g - ref32).max() < MEL_TOL and np.abs(g - ref64).max() < MEL_TOL, form
            if form == 0:
                assert np.abs(g - ref64).max() < 5e-5
            np.testing.assert_allclose(float(mean), ref64.mean(), atol=2e-6)
            assert float(stats[1]) == B * (1 + L // 160) * 80
            h = ops.logmel(m.plan(), x.half(), eps=m.eps, subtract_mean=False)
            assert torch.equal(h, ops.logmel(m.plan(), x.half().float(), eps=m.eps, subtract_mean=False))
    finally:
        N_.set_option("logmel_mfma", 0)


@pytest.mark.parametrize("wide", [False, True])
def test_logmel_custom_window_and_filterbank(wide):
    """The buffers a checkpoint may carry: a window that is NOT symmetric (the matrix form then takes its direct 400-term DFT,
    the fast form applies any window in the time domain) and a filterbank other than the HTK one -- `wide`: supports of 40 bins,
    which do not fit the fast form's compact filter table (weights then come from the [80][48] table in memory).  Both forms
    against a float64 restatement with the same buffers."""
    from tal_asrd_amd import LogMelSpec, synth, ops, _native as N_
    rng = np.random.default_rng(5 + int(wide))
    m = LogMelSpec().to(dev())
    window = (0.2 + 0.8 * np.sin(np.pi * np.arange(400) / 400.0) ** 2 * (1.0 + 0.3 * np.arange(400) / 400.0)).astype(np.float32)
    fb = np.zeros((201, 80), dtype=np.float32)
    width = 40 if wide else 6
    for j in range(80):
        lo = int(round(j * (201 - width) / 79.0))
        fb[lo:lo + width, j] = rng.uniform(0.1, 1.0, size=width).astype(np.float32)
    with torch.no_grad():
        m.mel_transform.spectrogram.window.copy_(torch.from_numpy(window))
        m.mel_transform.mel_scale.fb.copy_(torch.from_numpy(fb))
    audio = synth.synth_audio_batch(2, 20000, 77)
    x = torch.from_numpy(audio).to(dev())
    want = _logmel_f64(audio, window, fb, m.eps)
    try:
        for form in (0, 1):
            N_.set_option("logmel_mfma", form)
            got = ops.logmel(m.plan(), x, eps=m.eps, subtract_mean=False).cpu().numpy()
            np.testing.assert_allclose(got, want, atol=2e-5 if form == 0 else MEL_TOL, rtol=0, err_msg="form %d" % form)
    finally:
        N_.set_option("logmel_mfma", 0)


# ------------------------------------------------------------------ grouped convs
@pytest.mark.parametrize("cig,cog,T,B", [(1, 10, 333, 2), (10, 14, 300, 1), (14, 18, 277, 2), (2, 3, 64, 1)])
def test_gconv_s2_matches_torch(cig, cog, T, B):
    from tal_asrd_amd import ops
    G = 80 if cig != 2 else 8
    g = torch.Generator().manual_seed(cig * 100 + cog)
    x = torch.randn(B, G * cig, T, generator=g)
    w = torch.randn(G * cog, cig, 21, generator=g) / (21 * cig) ** 0.5
    b = torch.randn(G * cog, generator=g)
    ref = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=2, groups=G).permute(0, 2, 1)
    wp = ops.pack_gconv_weight(w.to(dev()), G)
    y = ops.gconv_s2(x.permute(0, 2, 1).contiguous().to(dev()), wp, b.to(dev()), G * cog, G)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cg,T,B", [(10, 700, 2), (14, 256, 1), (18, 300, 2), (18, 5, 1), (4, 50, 2)])
def test_gconv_res_matches_torch(cg, T, B):
    from tal_asrd_amd import ops
    G = 80 if cg != 4 else 8
    g = torch.Generator().manual_seed(cg)
    x = torch.randn(B, G * cg, T, generator=g)
    w = torch.randn(G * cg, cg, 21, generator=g) / (21 * cg) ** 0.5
    b = torch.randn(G * cg, generator=g)
    xd = x.double()
    ref = (xd + 0.25 * torch.relu(torch.nn.functional.conv1d(xd, w.double(), b.double(), padding=10, groups=G)))
    wp = ops.pack_gconv_weight(w.to(dev()), G)
    y = ops.gconv_res(x.permute(0, 2, 1).contiguous().to(dev()), wp, b.to(dev()), 0.25, G)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.permute(0, 2, 1).numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cg,T,B", [(10, 700, 2), (14, 256, 1), (14, 1000, 1), (18, 300, 2), (18, 5, 1), (10, 1, 1), (18, 529, 1)])
def test_gconv_res_f16x3_matches_float64(cg, T, B):
    """TDSBlock grouped conv on the matrix cores (fp16x3 form): same tolerance as the fp32 VALU kernel, the fused
    hi / lo split of the output is bit-identical to tal_split_f16x3_fwd of the fp32 output, rows behind the
    output stay untouched."""
    from tal_asrd_amd import ops
    G = 80
    g = torch.Generator().manual_seed(1000 + cg + T)
    x = torch.randn(B, G * cg, T, generator=g) * 2.0
    w = torch.randn(G * cg, cg, 21, generator=g) / (21 * cg) ** 0.5
    b = torch.randn(G * cg, generator=g)
    xd = x.double()
    ref = (xd + 0.25 * torch.relu(torch.nn.functional.conv1d(xd, w.double(), b.double(), padding=10, groups=G)))
    wf = ops.pack_gconv_f16x3_weight(w.to(dev()), G)
    assert wf is not None
    xt = x.permute(0, 2, 1).contiguous().to(dev())
    y, ys = ops.gconv_res_f16x3(xt, wf, b.to(dev()), 0.25, G, want_split=True)
    y2 = ops.gconv_res_f16x3(xt, wf, b.to(dev()), 0.25, G)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.permute(0, 2, 1).numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(y, y2)
    if (G * cg) % 32 == 0:
        want = ops.split_f16x3(y.reshape(B * T, G * cg))
        assert torch.equal(ys, want)


@pytest.mark.parametrize("T,B", [(1, 1), (7, 2), (64, 1), (65, 1), (129, 2), (255, 1), (256, 1), (300, 2), (1031, 1), (4000, 1)])
def test_gconv18_time_shift_packing(T, B):
    """18 channels per group, split form in and out (the six TDSBlocks of the last stage): the time-shift-packed kernel
    (channels 16, 17 of 128 steps as ONE shifted M tile) against the two-M-tile kernel it replaces -- channels 0-15 of every
    group are the same chains of MFMAs (bit-identical halves), channels 16, 17 agree with float64 to the conv tolerance; on
    both tile lengths (64 / 256 steps), which are bit-identical to each other."""
    from tal_asrd_amd import ops, _native as N_
    from tests.test_gpu_stress import _unsplit
    G, cg = 80, 18
    C = G * cg
    g = torch.Generator().manual_seed(4100 + T)
    x = torch.randn(B, T, C, generator=g) * 3.0
    w = torch.randn(C, cg, 21, generator=g) / (21 * cg) ** 0.5
    b = torch.randn(C, generator=g)
    xs = ops.split_f16x3(x.reshape(B * T, C).to(dev()))
    xq = _unsplit(xs, B * T, C).reshape(B, T, C).cpu()
    ref = xq + 0.4 * torch.relu(torch.nn.functional.conv1d(xq.permute(0, 2, 1), w.double(), b.double(), padding=10, groups=G)).permute(0, 2, 1)
    wf = ops.pack_gconv_f16x3_weight(w.to(dev()), G)
    out = {}
    try:
        for shift in (1, 0):
            for below in (0, 1 << 20):
                N_.set_option("gconv_no_shift18", 1 - shift)
                N_.set_option("gconv_short_below", below)
                out[(shift, below)] = ops.gconv_res_split(xs, (B, T, C), wf, b.to(dev()), 0.4, G).clone()
        torch.cuda.synchronize()
    finally:
        N_.set_option("gconv_no_shift18", 0)
        N_.set_option("gconv_short_below", 4)
    assert torch.equal(out[(1, 0)], out[(1, 1 << 20)])                      # long tiles == short tiles
    got_new = _unsplit(out[(1, 0)], B * T, C).reshape(B * T, G, cg)
    got_old = _unsplit(out[(0, 0)], B * T, C).reshape(B * T, G, cg)
    assert torch.equal(got_new[:, :, :16], got_old[:, :, :16])              # the 16-channel tile: identical chains
    np.testing.assert_allclose(got_new.reshape(B, T, C).cpu().numpy(), ref.numpy(), atol=3e-5, rtol=1e-5)
    assert float((got_new[:, :, 16:] - got_old[:, :, 16:]).abs().max()) < 2e-5


@pytest.mark.parametrize("cg,T,B", [(10, 1501, 1), (14, 751, 2), (18, 376, 1), (18, 65, 3)])
def test_gconv_short_tiles_equal_long_tiles(cg, T, B):
    """Short inputs take tiles of 64 output steps (option gconv_short_below): every output is the same chain of MFMAs as on the
    256-step (stride 2: 128-step) tiles, so fp32 and split-form results are BIT-IDENTICAL between the two."""
    from tal_asrd_amd import ops, _native as N_
    G = 80
    g = torch.Generator().manual_seed(77 + cg + T)
    x = (torch.randn(B, T, G * cg, generator=g) * 2.0).to(dev())
    w = (torch.randn(G * cg, cg, 21, generator=g) / (21 * cg) ** 0.5).to(dev())
    b = torch.randn(G * cg, generator=g).to(dev())
    wf = ops.pack_gconv_f16x3_weight(w, G)
    cog = {10: 14, 14: 18}.get(cg)
    if cog:
        w2 = (torch.randn(G * cog, cg, 21, generator=g) / (21 * cg) ** 0.5).to(dev())
        b2 = torch.randn(G * cog, generator=g).to(dev())
        wf2 = ops.pack_gconv_f16x3_weight(w2, G, stride=2)
    xs = ops.split_f16x3(x.reshape(B * T, G * cg)) if (G * cg) % 32 == 0 else None
    out = {}
    try:
        # (0: long tiles at their default lengths -- 256 steps, 128 for 14 channels per group since round 6 --; 1 << 20: 64-step tiles;
        #  then the long tiles forced to 256 and to 128 steps, option gconv_long_tt)
        for below, long_tt in ((0, 0), (1 << 20, 0), (0, 256), (0, 128)):
            N_.set_option("gconv_short_below", below)
            N_.set_option("gconv_long_tt", long_tt)
            r = [ops.gconv_res_f16x3(x, wf, b, 0.25, G)]
            if xs is not None:
                r.append(ops.gconv_res_split(xs, (B, T, G * cg), wf, b, 0.25, G))
            if cog and T >= 21:
                r.append(ops.gconv_s2_f16x3(x, wf2, b2, G * cog, G))
                if xs is not None and (G * cog) % 32 == 0:
                    r.append(ops.gconv_s2_split(xs, (B, T, G * cg), True, wf2, b2, G * cog, G))
            torch.cuda.synchronize()
            out[(below, long_tt)] = r
    finally:
        N_.set_option("gconv_short_below", 4)
        N_.set_option("gconv_long_tt", 0)
    ref = out[(0, 0)]
    assert len(ref) >= 2
    for key, res in out.items():
        assert len(res) == len(ref), key
        for a, c in zip(ref, res):
            assert torch.equal(a, c), key


@pytest.mark.parametrize("M", [1900, 3751])
def test_linear_f16x3_96_wide_tiles_equal_160_wide_tiles(M):
    """Mid-sized launches with N % 96 == 0 (a 5-minute clip's third stage) take 128 x 96 tiles when those fill the chip better than
    128 x 160 (option gemm_no_n96): the same MFMA chain per output, so relu layer and split-residual layer are BIT-IDENTICAL."""
    from tal_asrd_amd import ops, _native as N_
    lib = N_.lib()
    C = K = 1440
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(dev())
    w0 = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev())
    b0 = torch.randn(C, generator=g).to(dev())
    xs, w0s = ops.split_f16x3(x), ops.split_f16x3(w0)
    nws = lib.tal_linear_workspace_bytes(M, C, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev())
    flag = torch.zeros(16, dtype=torch.int32, device=dev())

    def run(mode):
        y = torch.full(((M + 2) * C * 4,), 0x5A, dtype=torch.uint8, device=dev())
        N_.check(lib.tal_linear_f16x3_guarded_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), N_.ptr(xs) if mode == 2 else None, 1 if mode == 2 else 0,
                                                  0.3, mode, M, C, K, N_.ptr(y), 1, N_.ptr(flag), N_.ptr(ws), nws, N_.stream_handle()),
                 "tal_linear_f16x3_guarded_fwd")
        torch.cuda.synchronize()
        assert bool((y[M * C * 4:] == 0x5A).all())
        return y
    try:
        N_.set_option("gemm_s64_below", 0)          # (1900 rows would otherwise take the 64 x 80 tiles)
        for mode in (1, 2):
            N_.set_option("gemm_no_n96", 0)
            a = run(mode)
            N_.set_option("gemm_no_n96", 1)
            assert torch.equal(a, run(mode)), mode
    finally:
        N_.set_option("gemm_no_n96", 0)
        N_.set_option("gemm_s64_below", 2)
    h = a[:M * C * 4].view(torch.float16).reshape(M, C // 32, 64).float()
    got = (h[:, :, :32] + h[:, :, 32:] / 2048.0).reshape(M, C).double()
    xd = ops.split_f16x3(x)[:M * K * 4].view(torch.float16).reshape(M, K // 32, 64).float()
    xd = (xd[:, :, :32] + xd[:, :, 32:] / 2048.0).reshape(M, K).double()
    ref = xd + 0.3 * (xd @ w0.double().t() + b0.double())
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, K ** 0.5 / 8)
    assert int(flag[0]) == 0


@pytest.mark.parametrize("M,C", [(376, 1440), (1501 + 3, 800), (751, 1120), (130, 800)])
def test_linear_f16x3_short_input_kernel(M, C):
    """The 64 x 80-tile kernel for short inputs (csrc/gemm_s64.hip; option gemm_s64_below) against float64 and against the
    K-sliced 128 x 160 launches it replaces: relu layer and residual layer, fp32 and split-form output, fp32 and split-form
    residual, guarded and unguarded; rows behind the output untouched; bitwise repeatable."""
    from tal_asrd_amd import ops, _native as N_
    lib = N_.lib()
    g = torch.Generator().manual_seed(M + C)
    K = C
    x = torch.randn(M, K, generator=g).to(dev())
    w0 = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev())
    b0 = torch.randn(C, generator=g).to(dev())
    xs, w0s = ops.split_f16x3(x), ops.split_f16x3(w0)
    nws = lib.tal_linear_workspace_bytes(M, C, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev())
    flag = torch.zeros(16, dtype=torch.int32, device=dev())
    guard = 3

    def run(mode, out_split, guarded):
        y = torch.full(((M + guard) * C * 4,), 0x5A, dtype=torch.uint8, device=dev())
        if guarded:
            N_.check(lib.tal_linear_f16x3_guarded_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), N_.ptr(xs) if mode == 2 else None, 1 if mode == 2 else 0,
                                                      0.3, mode, M, C, K, N_.ptr(y), out_split, N_.ptr(flag), N_.ptr(ws), nws, N_.stream_handle()),
                     "tal_linear_f16x3_guarded_fwd")
        else:
            N_.check(lib.tal_linear_f16x3_fwd(N_.ptr(xs), N_.ptr(w0s), N_.ptr(b0), N_.ptr(x) if mode == 2 else None, 0.3, mode, M, C, K, N_.ptr(y),
                                              out_split, N_.ptr(ws), nws, N_.stream_handle()), "tal_linear_f16x3_fwd")
        torch.cuda.synchronize()
        assert bool((y[M * C * 4:] == 0x5A).all())
        return y[:M * C * 4]

    def decode(y, out_split):
        if not out_split:
            return y.view(torch.float32).reshape(M, C)
        h = y.view(torch.float16).reshape(M, C // 32, 64).float()
        return (h[:, :, :32] + h[:, :, 32:] / 2048.0).reshape(M, C)

    xd = decode(xs.view(torch.uint8).reshape(-1), 1).double()         # what both kernels are given
    lin = xd @ w0.double().t() + b0.double()
    tol = 2e-5 * max(1.0, K ** 0.5 / 8)
    try:
        for mode, out_split, guarded in [(1, 0, False), (2, 0, False), (1, 1, False), (1, 1, True), (2, 1, True), (2, 0, True)]:
            N_.set_option("gemm_s64_below", 1 << 20)
            a = run(mode, out_split, guarded)
            assert torch.equal(a, run(mode, out_split, guarded))
            N_.set_option("gemm_s64_below", 0)
            b = run(mode, out_split, guarded)
            res = (xd if guarded else x.double())
            ref = torch.relu(lin) if mode == 1 else res + 0.3 * lin
            ea = float((decode(a, out_split).double() - ref).abs().max())
            eb = float((decode(b, out_split).double() - ref).abs().max())
            assert ea < tol and eb < tol, (mode, out_split, guarded, ea, eb)
            assert float((decode(a, out_split) - decode(b, out_split)).abs().max()) < 2e-5, (mode, out_split, guarded)
    finally:
        N_.set_option("gemm_s64_below", 2)
    assert int(flag[0]) == 0


@pytest.mark.parametrize("cig,cog,T,B", [(10, 14, 300, 1), (14, 18, 277, 2), (10, 14, 1000, 2), (14, 18, 21, 1), (14, 18, 22, 1),
                                        (10, 14, 555, 1)])
def test_gconv_s2_f16x3_matches_float64(cig, cog, T, B):
    """stride-2 resize conv on the matrix cores (even / odd input rows as two K segments): same tolerance as the fp32
    VALU kernel, rows behind the output untouched."""
    from tal_asrd_amd import ops
    G = 80
    g = torch.Generator().manual_seed(cig * 100 + cog + T)
    x = torch.randn(B, G * cig, T, generator=g) * 2.0
    w = torch.randn(G * cog, cig, 21, generator=g) / (21 * cig) ** 0.5
    b = torch.randn(G * cog, generator=g)
    ref = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=2, groups=G).permute(0, 2, 1)
    wf = ops.pack_gconv_f16x3_weight(w.to(dev()), G, stride=2)
    assert wf is not None
    y = ops.gconv_s2_f16x3(x.permute(0, 2, 1).contiguous().to(dev()), wf, b.to(dev()), G * cog, G)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


def test_gconv_f16x3_unsupported_width_reports_zero_bytes():
    from tal_asrd_amd import _native as N
    lib = N.lib()
    assert lib.tal_gconv_f16x3_weight_bytes(80 * 12, 80 * 12, 80, 1) == 0
    assert lib.tal_gconv_f16x3_weight_bytes(8 * 4, 8 * 4, 8, 1) == 0
    assert lib.tal_gconv_f16x3_weight_bytes(80, 800, 80, 2) == 0          # 1 -> 10 channels per group stays on the VALU
    assert lib.tal_gconv_f16x3_weight_bytes(800, 800, 80, 1) > 0 and lib.tal_gconv_f16x3_weight_bytes(1440, 1440, 80, 1) > 0
    assert lib.tal_gconv_f16x3_weight_bytes(800, 1120, 80, 2) > 0 and lib.tal_gconv_f16x3_weight_bytes(1120, 1440, 80, 2) > 0


# ------------------------------------------------------------------ golden: small TDS / block through the module API
def test_tds_small_golden():
    from tal_asrd_amd import TDS, synth
    g = golden("tds_small")
    m = TDS(input_size=8, sizes=[8, 16, 24, 32], depths=[1, 1, 2], kernel_size=21)
    sd = synth.fill_state_dict({"tds_small." + k: tuple(v.shape) for k, v in m.state_dict().items()})
    _load(m, {k[len("tds_small."):]: v for k, v in sd.items()})
    y = m(torch.from_numpy(g["x"]).to(dev()))
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], atol=2e-5, rtol=0)


def test_tdsblock_golden():
    from tal_asrd_amd import TDSBlock, synth
    g = golden("tdsblock")
    m = TDSBlock(32, 21, 8)
    sd = synth.fill_state_dict({"tdsblock." + k: tuple(v.shape) for k, v in m.state_dict().items()})
    _load(m, {k[len("tdsblock."):]: v for k, v in sd.items()})
    y = m(torch.from_numpy(g["x"]).to(dev()))
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], atol=2e-5, rtol=0)


# ------------------------------------------------------------------ golden: full SD path (configs 1 and 2)
def _check_sd_golden(model, name, with_lens):
    from tal_asrd_amd import synth
    g = golden(name)
    B, L = int(g["batch"]), int(g["audio_len"])
    lens = g["audio_lens"].tolist() if with_lens else None
    audio = torch.from_numpy(synth.synth_audio_batch(B, L, int(g["audio_seed"]), lens=lens)).to(dev())
    with torch.no_grad():
        mel = model.extract_features(audio)
        np.testing.assert_allclose(mel[:, g["mel_rows"]].cpu().numpy(), g["mel_sample"], atol=MEL_TOL, rtol=0)
        enc = model.encode_features(mel, None if lens is None else torch.tensor(lens))
        eo = enc["encoder_out"]
        np.testing.assert_allclose(eo[:, g["enc_rows"]].cpu().numpy(), g["enc_sample"], atol=LOGIT_TOL, rtol=0)
        np.testing.assert_allclose(eo.double().sum(dim=1).cpu().numpy(), g["enc_chan_sum"], atol=5e-2, rtol=1e-4)
        logits = model.decode(enc)
        np.testing.assert_allclose(logits[:, g["logit_rows"]].cpu().numpy(), g["logit_sample"], atol=LOGIT_TOL, rtol=0)
        np.testing.assert_allclose(logits.max(-1).values.cpu().numpy(), g["logit_max"], atol=LOGIT_TOL, rtol=0)
        feat = model.spk_embed_proj(eo)
        if "feat" in g.files and g["feat"].shape[1] == feat.shape[1]:
            np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], atol=LOGIT_TOL, rtol=0)
        from tal_asrd_amd import ops
        ids = ops.argmax_rows(logits).cpu().numpy()
        np.testing.assert_array_equal(ids, logits.argmax(-1).cpu().numpy())   # argmax kernel == torch.argmax
        # committed fixtures: IDENTICAL speaker ids (their closest top-2 margin is far above fp32 noise; the
        # near-tie escape hatch lives only in the random-data tests)
        np.testing.assert_array_equal(ids, g["ids"])
        if with_lens:
            np.testing.assert_array_equal(enc["encoder_padding_mask"].cpu().numpy(), g["mask"])
    return float(np.abs(logits[:, g["logit_rows"]].cpu().numpy() - g["logit_sample"]).max())


def test_sd_30s_golden(sd_model):
    err = _check_sd_golden(sd_model, "sd_30s", False)
    print("sd_30s max logit err %.3e" % err)


def test_sd_b2_ragged_golden(sd_model):
    _check_sd_golden(sd_model, "sd_b2_ragged", True)


def test_sd_5min_golden(sd_model):
    _check_sd_golden(sd_model, "sd_5min", False)


def test_sd_b4_golden_one_call(sd_model):
    """ONE reference SDModel call on four 60 s segments (fixture recorded from the reference's own module): module API."""
    _check_sd_golden(sd_model, "sd_b4_60s", False)


def test_sd_b4_split_over_two_ranks_with_the_combined_mean(sd_model):
    """What each rank of BASELINE configs[3] does (bench.py `--workload segments`), pinned to the REFERENCE: the four segments of one
    reference call (`tal/asr/models.py:52` subtracts the mean of the whole call) are computed as 2 + 2 -- each half's log-mel
    without the mean and its (sum, count), the two (sum, count) pairs added as the scalar all-reduce adds them
    (distributed.allreduce_logmel_stats), `SDModel.speaker_ids_from_logmel(mel_half, call_mean)` -- and as one 4-segment call with
    its own statistics.  Speaker ids identical to the reference's on all 4 x 733 frames, features within 1e-3."""
    from tal_asrd_amd import distributed as D, ops, synth
    g = golden("sd_b4_60s")
    B, L = int(g["batch"]), int(g["audio_len"])
    assert B == 4
    audio = torch.from_numpy(synth.synth_audio_batch(B, L, int(g["audio_seed"]))).to(dev())
    lm = sd_model.logmelspec
    with torch.no_grad():
        halves = [audio[0:2].contiguous(), audio[2:4].contiguous()]
        parts = [ops.logmel(lm.plan(), h, eps=lm.eps, subtract_mean=False, return_stats=True) for h in halves]
        st = parts[0][2] + parts[1][2]                      # the all-reduce over the two ranks
        mean = D.allreduce_logmel_stats(st.clone())         # (one process: returns sum / count as the float32 scalar)
        # the combined mean is the mean of the one-call log-mel
        mel4, mean4, st4 = ops.logmel(lm.plan(), audio, eps=lm.eps, subtract_mean=False, return_stats=True)
        assert float(st4[1]) == float(st[1])
        assert abs(float(mean) - float(mean4)) <= 1e-6
        # a half's own mean is a different number: the test would not notice a missing all-reduce otherwise
        assert abs(float(parts[0][1]) - float(mean)) > 1e-4 or abs(float(parts[1][1]) - float(mean)) > 1e-4
        feats, idss = [], []
        for mel_h, _, _ in parts:
            f, i = sd_model.speaker_ids_from_logmel(mel_h, mean)
            feats.append(f)
            idss.append(i)
        feat, ids = torch.cat(feats, 0), torch.cat(idss, 0)
        np.testing.assert_array_equal(ids.cpu().numpy(), g["ids"])
        np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], atol=LOGIT_TOL, rtol=0)
        # the same call in one piece
        f1, i1 = sd_model.speaker_ids_from_logmel(mel4, mean4)
        np.testing.assert_array_equal(i1.cpu().numpy(), g["ids"])
        np.testing.assert_allclose(f1.cpu().numpy(), g["feat"], atol=LOGIT_TOL, rtol=0)
        # and with each half's OWN mean at least the features move (the coupling is real on this fixture)
        f_own, _ = sd_model.speaker_ids_from_logmel(parts[0][0], parts[0][1])
        assert float((f_own - feats[0]).abs().max()) > 1e-5


@pytest.mark.parametrize("B,seconds", [(1, 30), (1, 300), (2, 47), (1, 1200), (8, 300)])
def test_first_resize_conv_inside_the_first_block_conv_launch(sd_model, B, seconds):
    """Round 6, option gconv_c1_fuse (measured no faster, off by default): the encoder's first resize conv (1 mel bin -> 10 channels per
    group, tal/asr/models.py:363-364) computed straight into the LDS slab of the first TDSBlock conv's launch (gconv_mfma_kernel<..,
    FROMC1>: same fmaf chain, mean fold and hi / lo split as gconv_s2_c1_kernel) -- the stage's first activation never exists in
    memory.  Against the two launches of the default path: features and speaker ids BIT-identical, with and without the folded mean, short (64-step) and long
    (256-step) tiles, a batch, a length whose last tile is partial."""
    from tal_asrd_amd import _native as N, ops, synth
    L = seconds * 16000 + 1234
    audio = torch.from_numpy(synth.synth_audio_batch(B, L, 321)).to(dev())
    enc = sd_model.encoder
    out = {}
    try:
        for nofuse in (0, 1):
            N.set_option("gconv_c1_fuse", 1 - nofuse)
            with torch.no_grad():
                f, i = sd_model.speaker_ids(audio)                   # the mean rides in the first conv's bias
                mel = sd_model.extract_features(audio)               # the mean already subtracted
                y = ops.tds_forward(enc._descriptor(0, len(enc.sizes) - 1), mel, enc.sizes[-1])
            out[nofuse] = (f.clone(), i.clone(), y.clone())
    finally:
        N.set_option("gconv_c1_fuse", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][2], out[1][2])


def test_speaker_ids_fused_path(sd_model):
    """reconcile.get_speaker_ids form: ids without materialised logits == ids with logits."""
    from tal_asrd_amd import synth
    g = golden("sd_30s")
    audio = torch.from_numpy(synth.synth_audio_batch(1, 480000, 1234)).to(dev())
    feat, ids = sd_model.speaker_ids(audio)
    feat2, ids2, logits = sd_model.speaker_ids(audio, want_logits=True)
    np.testing.assert_array_equal(ids.cpu().numpy(), ids2.cpu().numpy())
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], atol=LOGIT_TOL, rtol=0)
    assert (ids.cpu().numpy() != g["ids"]).sum() == 0


@pytest.mark.parametrize("name", ["sd_30s", "sd_5min", "sd_b2_ragged"])
def test_mean_folded_into_the_first_resize_conv(sd_model, name):
    """LogMelSpec subtracts the global mean of its call (tal/asr/models.py:52); SDModel.speaker_ids hands the UNSUBTRACTED log-mel
    and the scalar to tal_tds_premean_fwd, where it becomes a bias correction of the first (padding-free) resize conv.  Against the
    call sequence with the separate subtraction pass (extract_features -> tal_tds_fwd -> head): encoder output and features equal
    to fp32 rounding of that one conv, speaker ids identical and equal to the reference's; extract_features itself is unchanged;
    a stack whose first conv is not the 1 -> 10 form is refused by the C entry and served by the subtraction pass in Python."""
    import ctypes as C
    from tal_asrd_amd import synth, ops, _native as N
    g = golden(name)
    B, L = int(g["batch"]), int(g["audio_len"])
    lens = g["audio_lens"].tolist() if "audio_lens" in g.files and name == "sd_b2_ragged" else None
    audio = torch.from_numpy(synth.synth_audio_batch(B, L, int(g["audio_seed"]), lens=lens)).to(dev())
    enc = sd_model.encoder
    heads = (sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias, sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias)
    with torch.no_grad():
        mel = sd_model.extract_features(audio)
        raw, mean = sd_model.logmelspec.forward_unsubtracted(audio)
        np.testing.assert_array_equal((raw - mean).cpu().numpy(), mel.cpu().numpy())       # the same tensor, one subtraction apart
        desc = enc._descriptor(0, len(enc.sizes) - 1)
        assert ops.tds_premean_ok(desc, raw)
        y_sub = ops.tds_forward(desc, mel, enc.sizes[-1])
        y_fold = ops.tds_forward(desc, raw, enc.sizes[-1], x_mean=mean)
        np.testing.assert_allclose(y_fold.cpu().numpy(), y_sub.cpu().numpy(), atol=2e-4, rtol=0)
        f_sub, _, i_sub = ops.sd_head(y_sub, *heads, want_logits=False, want_ids=True)
        feat, ids = sd_model.speaker_ids(audio)
        np.testing.assert_allclose(feat.cpu().numpy(), f_sub.cpu().numpy(), atol=1e-4, rtol=0)
        np.testing.assert_array_equal(ids.cpu().numpy(), i_sub.cpu().numpy())
        np.testing.assert_array_equal(ids.cpu().numpy(), g["ids"])
        # exact-fp32 descriptor: the same entry
        exact = N.TdsDesc.from_buffer_copy(desc)
        exact.flags |= N.TAL_TDS_EXACT_F32
        y_ex = ops.tds_forward(exact, raw, enc.sizes[-1], x_mean=mean)
        np.testing.assert_allclose(y_ex.cpu().numpy(), y_sub.cpu().numpy(), atol=2e-4, rtol=0)
        # a stack that starts with the 10 -> 14 conv has no mean-folding kernel: refused with a message, nothing launched
        sub = enc._descriptor(1, len(enc.sizes) - 1)
        x1 = torch.zeros(1, 400, enc.sizes[1], device=dev())
        assert not ops.tds_premean_ok(sub, x1)
        lib = N.lib()
        nws = lib.tal_tds_workspace_bytes(C.byref(sub), 1, 400)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev())
        out = torch.empty(1, lib.tal_tds_out_len(C.byref(sub), 400), enc.sizes[-1], device=dev())
        rc = lib.tal_tds_premean_fwd(C.byref(sub), N.ptr(x1), N.ptr(mean), 1, 400, N.ptr(out), N.ptr(ws), nws, N.stream_handle())
        assert rc == -1 and b"mean-folding" in lib.tal_last_error()


@pytest.mark.parametrize("name", ["sd_30s", "sd_5min", "sd_b2_ragged"])
def test_head_on_the_split_encoder_output(sd_model, name):
    """SDModel.speaker_ids leaves the encoder output of a long input in the hi / lo split form (TAL_TDS_OUT_SPLIT: the last dense
    layer writes no fp32 copy) and runs the head's 1440 -> 128 embedding layer on it in the fp16x3 form (tal_sd_head_split_fwd).
    Against the fp32 output + fp32 embedding layer: the split tensor decodes to the fp32 one within the form's 22 mantissa bits,
    features within 1e-4, speaker ids identical and equal to the reference's; with logits requested the same; an exact-mode
    descriptor never yields the split form."""
    import ctypes as C
    from tal_asrd_amd import synth, ops, _native as N
    g = golden(name)
    B, L = int(g["batch"]), int(g["audio_len"])
    lens = g["audio_lens"].tolist() if name == "sd_b2_ragged" else None
    audio = torch.from_numpy(synth.synth_audio_batch(B, L, int(g["audio_seed"]), lens=lens)).to(dev())
    enc = sd_model.encoder
    heads = (sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias, sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias)
    with torch.no_grad():
        mel = sd_model.extract_features(audio)
        desc = enc._descriptor(0, len(enc.sizes) - 1)
        y32 = ops.tds_forward(desc, mel, enc.sizes[-1])
        ysp, chk = ops.tds_forward(desc, mel, enc.sizes[-1], defer=True, out_split=True)
        assert chk.y_split and not chk.flagged()
        Cc = enc.sizes[-1]
        h = ysp.view(torch.float16).view(-1, Cc // 32, 2, 32).float()
        dec = (h[:, :, 0] + h[:, :, 1] / 2048.0).reshape(y32.shape)
        np.testing.assert_allclose(dec.cpu().numpy(), y32.cpu().numpy(), atol=2e-6 * float(y32.abs().max()) + 1e-7, rtol=0)
        f32, l32, i32 = ops.sd_head(y32, *heads, want_logits=True, want_ids=True)
        fsp, lsp, isp = ops.sd_head(ysp, *heads, want_logits=True, want_ids=True, x_split=True, w_embed_split=sd_model._embed_split())
        np.testing.assert_allclose(fsp.cpu().numpy(), f32.cpu().numpy(), atol=1e-4, rtol=0)
        np.testing.assert_allclose(lsp.cpu().numpy(), l32.cpu().numpy(), atol=LOGIT_TOL, rtol=0)
        np.testing.assert_array_equal(isp.cpu().numpy(), i32.cpu().numpy())
        _, _, ifast = ops.sd_head(ysp, *heads, want_logits=False, want_ids=True, x_split=True, w_embed_split=sd_model._embed_split())
        np.testing.assert_array_equal(ifast.cpu().numpy(), g["ids"])
        feat, ids, logits = sd_model.speaker_ids(audio, want_logits=True)
        np.testing.assert_array_equal(ids.cpu().numpy(), g["ids"])
        np.testing.assert_allclose(logits[:, g["logit_rows"]].cpu().numpy(), g["logit_sample"], atol=LOGIT_TOL, rtol=0)
        exact = N.TdsDesc.from_buffer_copy(desc)
        exact.flags |= N.TAL_TDS_EXACT_F32 | N.TAL_TDS_OUT_SPLIT
        assert N.lib().tal_tds_out_split(C.byref(exact), B, mel.shape[1]) == 0
        _, chk2 = ops.tds_forward(exact, mel, enc.sizes[-1], defer=True, out_split=True)
        assert not chk2.y_split


def test_an_embedding_weight_outside_the_fp16_range_keeps_the_fp32_head(sd_model):
    """SDModel._embed_split() is None for an embedding weight the hi / lo split cannot carry (|w| > 65504): speaker_ids then
    asks for the fp32 encoder output and runs the fp32 embedding layer -- same ids and features as ops.sd_head on the fp32 path;
    the cached split follows the parameter's version (restoring the weight brings the split head back)."""
    from tal_asrd_amd import synth, ops
    audio = torch.from_numpy(synth.synth_audio_batch(1, 160000, 77)).to(dev())
    w = sd_model.spk_embed_proj.weight
    keep = w.detach().clone()
    with torch.no_grad():
        feat0, ids0 = sd_model.speaker_ids(audio)
        assert sd_model._embed_split() is not None
        try:
            w[0, 0] = 1.0e5
            assert sd_model._embed_split() is None
            feat, ids = sd_model.speaker_ids(audio)
            mel = sd_model.extract_features(audio)
            enc = sd_model.encode_features(mel, None)["encoder_out"]
            f32, _, i32 = ops.sd_head(enc, w, sd_model.spk_embed_proj.bias, sd_model.spk_logit_proj.weight,
                                      sd_model.spk_logit_proj.bias, want_logits=False, want_ids=True)
            np.testing.assert_array_equal(ids.cpu().numpy(), i32.cpu().numpy())
            np.testing.assert_allclose(feat.cpu().numpy(), f32.cpu().numpy(), atol=1e-4 * float(f32.abs().max()), rtol=0)
        finally:
            w.copy_(keep)
        assert sd_model._embed_split() is not None
        feat1, ids1 = sd_model.speaker_ids(audio)
        assert torch.equal(ids1, ids0) and torch.equal(feat1, feat0)


@pytest.mark.parametrize("seconds", [12, 300])
def test_the_sd_path_can_be_captured_as_a_hip_graph(sd_model, seconds):
    """The C-ABI calls of the SD path (log-mel, tal_tds_fwd, the head) issue nothing but kernel launches on the caller's stream,
    so a caller may capture them into a HIP graph and replay it: same features and ids as the eager call for other clips of
    the shape, and the fp16-range status block is cleared by every replay (it is cleared by a kernel: a captured
    hipMemsetAsync node replayed with a stale fill pattern on ROCm 7.2 -- profiles/r4_short_clip_graph.txt)."""
    from tal_asrd_amd import ops, synth
    enc = sd_model.encoder
    heads = (sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias, sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias)
    clips = [torch.from_numpy(synth.synth_audio_batch(1, seconds * 16000, 900 + k)).to(dev()) for k in range(3)]
    with torch.no_grad():
        sd_model.speaker_ids(clips[0])                      # everything that is built once (plans, packs, kernel attributes)
        torch.cuda.synchronize()
        static_x = clips[0].clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            mel, mean = sd_model.logmelspec.forward_unsubtracted(static_x)       # (speaker_ids' own sequence: the mean rides in the first conv's bias)
            y, chk = ops.tds_forward(enc._descriptor(0, len(enc.sizes) - 1), mel, enc.sizes[-1], defer=True, x_mean=mean, out_split=True)
            feat, _, ids = ops.sd_head(y, *heads, want_logits=False, want_ids=True, x_split=chk.y_split, w_embed_split=sd_model._embed_split())
        for a in clips + clips[:1]:
            for _ in range(20):
                f0, i0 = sd_model.speaker_ids(a)            # eager traffic between replays (what exposed the stale memset node)
            static_x.copy_(a)
            graph.replay()
            assert not chk.flagged()
            assert torch.equal(i0, ids) and torch.equal(f0, feat)


def test_speaker_ids_stream_over_clips_of_different_lengths(sd_model):
    """The episode-after-episode loop (tal/baseline/reconcile.py:96-102) over host clips that all differ in length (as real
    episodes do), one of them half precision: every result equals the one-clip call, in order, across two calls -- and the
    model keeps exactly two upload buffers, sized to the longest clip, however many lengths it has seen."""
    from tal_asrd_amd import synth
    secs = [20, 45, 12, 33, 45, 8]
    clips = [torch.from_numpy(synth.synth_audio_batch(1, s * 16000 + 17 * i, 700 + i)).pin_memory() for i, s in enumerate(secs)]
    clips[3] = clips[3].half().pin_memory()
    want = []
    for c in clips:
        f, i = sd_model.speaker_ids(c.to(dev()))
        want.append((f.clone(), i.clone()))
    for rep in range(2):
        got = [(f.clone(), i.clone()) for f, i in sd_model.speaker_ids_stream(clips)]
        assert len(got) == len(clips)
        for (f, i), (wf, wi) in zip(got, want):
            assert torch.equal(i, wi) and torch.equal(f, wf)
    bufs = sd_model._stream_bufs
    assert len(bufs) == 2 and all(b is not None for b in bufs)
    longest = max(c.numel() * c.element_size() for c in clips)
    assert all(b.numel() <= longest + longest // 8 for b in bufs)
    assert list(sd_model.speaker_ids_stream([])) == []


@pytest.mark.parametrize("seconds", [12, 30, 95, 300, 420])
def test_speaker_ids_do_not_depend_on_dispatch_choices(sd_model, seconds):
    """Which kernels a launch takes depends on its size (64 x 80 / 128 x 96 / 128 x 160 / 256 x 160 dense tiles with or without K
    slices, a round + remainder split by rows, 64- / 256-step conv tiles, the K-sliced head layer): the same clip with every
    round-3 choice switched off must give the SAME speaker ids and features equal to fp32 rounding."""
    from tal_asrd_amd import synth, _native as N_
    audio = torch.from_numpy(synth.synth_audio_batch(1, seconds * 16000, 4321 + seconds)).to(dev())
    names = {"gemm_s64_below": (2, 0), "gconv_short_below": (4, 0), "gemm_no_n96": (0, 1), "gemm_no_row_split": (0, 1), "gemm_no_w64": (0, 1),
             "head_no_astationary": (0, 1)}
    feat, ids = sd_model.speaker_ids(audio)
    torch.cuda.synchronize()
    try:
        for name, (on, off) in names.items():
            N_.set_option(name, off)
        feat2, ids2 = sd_model.speaker_ids(audio)
        torch.cuda.synchronize()
    finally:
        for name, (on, off) in names.items():
            N_.set_option(name, on)
    assert torch.equal(ids, ids2)
    assert float((feat - feat2).abs().max()) < 2e-4 * max(1.0, float(feat.abs().max()))
    # round 4: tile order and 32-row tiles of the short-input dense layer change which workgroup computes a tile, not its arithmetic
    try:
        N_.set_option("gemm_s64_order", 1)
        N_.set_option("gemm_s64_rows", 1)
        feat3, ids3 = sd_model.speaker_ids(audio)
        N_.set_option("gemm_s64_order", 2)
        N_.set_option("gemm_s64_rows", 2)
        feat4, ids4 = sd_model.speaker_ids(audio)
        torch.cuda.synchronize()
    finally:
        N_.set_option("gemm_s64_order", 0)
        N_.set_option("gemm_s64_rows", 0)
    assert torch.equal(ids, ids3) and torch.equal(feat, feat3) and torch.equal(ids, ids4) and torch.equal(feat, feat4)


@pytest.mark.parametrize("M,S", [(33000, 6008), (32768 + 1, 6008), (70001, 1000), (29864, 6008), (16800, 6008), (17001, 6008),
                                 (9000, 6008), (3751, 6008), (751, 6008), (376, 6008), (300, 6008), (1100, 1000)])
def test_long_input_argmax_head_matches_logits_argmax(M, S):
    """The A-stationary arg-max kernel (128-d features; 3 partial slots per row on long inputs, up to one per N tile on clips of
    seconds to minutes): same ids as arg-max over materialised
    logits, ragged last row block and last column tile, and the LOWEST index on exact ties (two identical
    speaker rows made the winners of a band of rows), as torch.argmax does."""
    from tal_asrd_amd import ops
    g = torch.Generator().manual_seed(M + S)
    x = torch.randn(M, 64, generator=g)
    we = torch.randn(128, 64, generator=g) / 8
    be = torch.randn(128, generator=g)
    wl = torch.randn(S, 128, generator=g) / 11
    bl = torch.randn(S, generator=g)
    lo, hi = 37, S - 5                       # a tie between a column of the first tile and one of the last
    wl[hi] = wl[lo]
    bl[lo] = bl[hi] = 50.0                   # large bias: the two tie for the maximum of every row
    d = dev()
    feat, logits, ids_ref = ops.sd_head(x.to(d), we.to(d), be.to(d), wl.to(d), bl.to(d), want_logits=True, want_ids=True)
    feat2, _, ids = ops.sd_head(x.to(d), we.to(d), be.to(d), wl.to(d), bl.to(d), want_logits=False, want_ids=True)
    torch.cuda.synchronize()
    assert torch.equal(feat, feat2)
    assert torch.equal(ids, ids_ref)
    assert torch.equal(ids.long().cpu(), logits.argmax(-1).cpu())
    assert bool((ids == lo).all())
    # without the planted winners: ordinary data
    bl[lo] = bl[hi] = 0.0
    _, logits, ids_ref = ops.sd_head(x.to(d), we.to(d), be.to(d), wl.to(d), bl.to(d), want_logits=True, want_ids=True)
    _, _, ids = ops.sd_head(x.to(d), we.to(d), be.to(d), wl.to(d), bl.to(d), want_logits=False, want_ids=True)
    assert torch.equal(ids.long().cpu(), logits.argmax(-1).cpu())


def test_asr_encode_golden(asr_model):
    from tal_asrd_amd import synth
    g = golden("asr_enc_b2")
    lens = g["audio_lens"].tolist()
    audio = torch.from_numpy(synth.synth_audio_batch(2, 480000, 1234, lens=lens)).to(dev())
    with torch.no_grad():
        enc = asr_model.encode(audio, torch.tensor(lens))
    r = g["rows"]
    np.testing.assert_allclose(enc["encoder_out"][:, r].cpu().numpy(), g["encoder_out"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(enc["speaker_out"][:, r].cpu().numpy(), g["speaker_out"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_array_equal(enc["encoder_padding_mask"].cpu().numpy(), g["mask"])


# ------------------------------------------------------------------ size-independent properties at full size
def test_time_shift_equivariance_full_size(sd_model):
    """Interior encoder frames depend only on their receptive field (output c <- mel frames
    [8c-640, 8c+780]): shifting the mel input by 8 frames shifts the output by 1."""
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(1, 30001, 80, generator=g).to(dev())
    a = sd_model.encoder.forward_time_major(mel)
    b = sd_model.encoder.forward_time_major(mel[:, 8:].contiguous())
    lo, hi = 100, a.shape[1] - 120
    np.testing.assert_allclose(a[0, lo + 1:hi + 1].cpu().numpy(), b[0, lo:hi].cpu().numpy(), atol=2e-4, rtol=0)


def test_batch_items_independent(sd_model):
    """Encoder output of a batch item does not depend on its neighbours (zero padding is per item)."""
    g = torch.Generator().manual_seed(6)
    mel = torch.randn(3, 2000, 80, generator=g).to(dev())
    full = sd_model.encoder.forward_time_major(mel)
    one = sd_model.encoder.forward_time_major(mel[1:2].contiguous())
    # kernel selection depends on the row count (M <= 512 rows take the split-K latency kernel, whose
    # summation order differs), so equality is to fp32 round-off, not bitwise
    np.testing.assert_allclose(full[1].cpu().numpy(), one[0].cpu().numpy(), atol=1e-5, rtol=0)


# ------------------------------------------------------------------ full-size (1 hour) properties
def test_one_hour_logmel_against_float64_oracle(sd_model):
    """configs[2] front-end at full size: LogMelSpec.forward on the 57.6 M-sample clip, 256 sampled rows and the global
    mean against the float64 restatement of torchaudio 0.4.0 (oracle.logmel_f64 on +-600-sample neighbourhoods of
    the sampled frames; the mean from float64 partial sums over the whole clip in 30-second pieces)."""
    from oracle import tal_oracle as O
    from tal_asrd_amd import ops, synth
    L = 57_600_000
    audio = synth.synth_audio_batch(1, L, 1234)
    x = torch.from_numpy(audio).to(dev())
    mel, mean, stats = ops.logmel(sd_model.logmelspec.plan(), x, eps=sd_model.logmelspec.eps, subtract_mean=True,
                                  return_stats=True)
    T = 1 + L // 160
    assert tuple(mel.shape) == (1, T, 80)
    # global mean: float64 reference, piecewise (each piece extended by the 200-sample reflect / overlap context)
    total = 0.0
    piece = 3000                                    # frames per piece
    for f0 in range(0, T, piece):
        f1 = min(T, f0 + piece)
        total += float(O.logmel_f64_frames(audio[0], f0, f1).sum())
    want_mean = total / (T * 80)
    assert abs(float(mean) - want_mean) < 2e-5, (float(mean), want_mean)
    assert abs(float(stats[0]) / float(stats[1]) - want_mean) < 2e-6
    rows = np.unique(np.concatenate([np.arange(0, 8), np.arange(T - 8, T), np.linspace(0, T - 1, 240).astype(np.int64)]))
    got = mel[0, torch.from_numpy(rows).to(dev())].cpu().numpy()
    want = np.stack([O.logmel_f64_frames(audio[0], int(r), int(r) + 1)[0] for r in rows]) - want_mean
    np.testing.assert_allclose(got, want, atol=MEL_TOL, rtol=0)


def test_one_hour_prefix_consistency_and_determinism(sd_model):
    """BASELINE.json's full size (360,001 mel frames): encoder frames whose receptive field
    (mel frames [8c-640, 8c+780]) lies inside a 5-minute prefix must equal the 5-minute run, and
    two runs of the 1-hour call are bit-identical (no atomics / order-dependent reductions)."""
    g = torch.Generator().manual_seed(8)
    mel = torch.randn(1, 360001, 80, generator=g).to(dev())
    a = sd_model.encoder.forward_time_major(mel)
    assert a.shape[1] == 44983
    b = sd_model.encoder.forward_time_major(mel)
    assert torch.equal(a, b)
    p = sd_model.encoder.forward_time_major(mel[:, :30001].contiguous())
    np.testing.assert_allclose(a[0, :3600].cpu().numpy(), p[0, :3600].cpu().numpy(), atol=1e-5, rtol=0)
    feat, ids = None, None
    from tal_asrd_amd import ops
    f1, l1, i1 = ops.sd_head(a[:, :4096].contiguous(), sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias,
                             sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias, True, True)
    f2, _, i2 = ops.sd_head(a, sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias,
                            sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias, False, True)
    # fused arg-max (no logits) == arg-max of materialised logits, at full size
    np.testing.assert_array_equal(i1[0].cpu().numpy(), i2[0, :4096].cpu().numpy())
    np.testing.assert_array_equal(l1.argmax(-1).cpu().numpy(), i1.cpu().numpy())


def test_head_argmax_is_the_same_call_after_call(sd_model):
    """The arg-max head on the row count of BASELINE.json configs[3] (64 x 3,733 rows), 40 calls on one input: ids and features of
    every call equal the first call's.  Round 5 found a race here (csrc/head.hip: a barrier with no wait for the LDS-DMA loads in
    front of it -- a few wrong ids in 14-100 % of the calls, profiles/r5_head_lds_dma_race.txt); the ids are also checked against
    the float64 arg-max of the logits on sampled rows whose margin is not within round-off."""
    from tal_asrd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.randn(64, 3733, 1440, generator=g, device=dev())
    heads = (sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias, sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias)
    with torch.no_grad():
        f0, _, i0 = ops.sd_head(x, *heads, False, True)
        for k in range(40):
            f, _, i = ops.sd_head(x, *heads, False, True)
            assert torch.equal(i, i0), "call %d: %d ids differ" % (k, int((i != i0).sum()))
            assert torch.equal(f, f0)
        rows = torch.randint(0, 64 * 3733, (4096,), generator=torch.Generator().manual_seed(1)).to(dev())
        feat = f0.reshape(-1, 128)[rows].double()
        logits = feat @ heads[2].double().t() + heads[3].double()
        top2 = torch.topk(logits, 2, dim=-1)
        safe = (top2.values[:, 0] - top2.values[:, 1]) > 1e-4
        assert bool((i0.reshape(-1)[rows][safe].long() == top2.indices[:, 0][safe]).all())


def test_config3_batch_shape_items_match_single_calls(sd_model):
    """BASELINE.json configs[3] shape on one GPU: a batch of 64 five-minute segments through the
    encoder + fused head in ONE call; any item must equal its own B=1 call (round-off only: the
    per-item zero padding keeps items independent, kernel selection may differ with the row count)."""
    from tal_asrd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    mel = torch.randn(64, 30001, 80, generator=g, device=dev())
    enc = sd_model.encoder.forward_time_major(mel)
    assert tuple(enc.shape) == (64, 3733, 1440)
    feat, _, ids = ops.sd_head(enc, sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias,
                               sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias, False, True)
    for i in (0, 17, 63):
        one = sd_model.encoder.forward_time_major(mel[i:i + 1].contiguous())
        np.testing.assert_allclose(enc[i].cpu().numpy(), one[0].cpu().numpy(), atol=1e-5, rtol=0)
        f1, l1, i1 = ops.sd_head(one, sd_model.spk_embed_proj.weight, sd_model.spk_embed_proj.bias,
                                 sd_model.spk_logit_proj.weight, sd_model.spk_logit_proj.bias, True, True)
        top2 = torch.topk(l1[0], 2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert bool((ids[i][safe] == i1[0][safe]).all())
