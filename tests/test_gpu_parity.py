"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle on the same
seeded inputs (and against the committed Hugging Face fixtures)."""
import numpy as np
import pytest

from handy_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wr():
    from handy_amd import whisper_rs
    whisper_rs.load_library()
    return whisper_rs


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [(200, 192, 256), (64, 64, 64), (1, 128, 128), (300, 256, 1280), (5, 51968 // 4, 128), (3000, 2048, 256), (2047, 1280, 1280),
                                   # two or more 256 x 256 tiles per CU: the persistent kernel (whole and ragged last row tile, fp32 rows out)
                                   (33024, 1280, 256), (26500, 1536, 384)])
def test_gemm_matches_numpy(wr, M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K), dtype=np.float32)
    W = rng.standard_normal((N, K), dtype=np.float32) + np.arange(N, dtype=np.float32)[:, None] * 0.01   # asymmetric
    C = np.empty((M, N), np.float32)
    L = wr.load_library()
    assert L.wmi_debug_gemm(M, N, K, wr._fp(A), wr._fp(W), wr._fp(C)) == 0
    ref = _f16(A).astype(np.float64) @ _f16(W).astype(np.float64).T
    err = np.abs(C - ref).max()
    assert err <= 2e-3 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("M", [1024, 24576, 132096])
def test_gelu_epilogue_over_every_f16_input(wr, M):
    """The GELU fused into the fc1 GEMM (every tile path: 64x64 for 1024 rows, 256x256 with the staged epilogue for 24576, the
    persistent 256x256 kernel -- 516 tiles, K = 128 -- for 132096) on every finite fp16 value,
    against ggml's definition -- the tanh form evaluated on the fp16 input and rounded to fp16 (the values of its lookup
    table), 0 at and below -10, the input itself at and above 10. The kernel's exp and reciprocal are hardware
    approximations: at most one fp16 ulp off (5e-7 absolute in the cancelling negative tail), on a small share of inputs."""
    import ctypes as C
    L = wr.load_library()
    bits = np.arange(65536, dtype=np.uint16)
    h = bits.view(np.float16)
    vals = np.where(np.isfinite(h), h, np.float16(0)).astype(np.float32)
    K = 128 if M > 100000 else 64
    A = np.zeros((M, K), np.float32)
    n_per = 65536 // 1024
    A[:1024, :n_per] = vals.reshape(1024, n_per)                           # value j of row i sits in column j
    A[1024:] = A[:1024].repeat(1, axis=0)[np.arange(M - 1024) % 1024] if M > 1024 else 0
    N = 256
    W = np.zeros((N, K), np.float32); W[np.arange(K), np.arange(K)] = 1.0   # identity: out[:, :64] = A
    out = np.empty((M, N), np.uint16)
    assert L.wmi_debug_gemm_gelu(M, N, K, wr._fp(A), wr._fp(W), None, out.ctypes.data_as(C.c_void_p)) == 0
    for r0 in range(0, M, 1024):
        got = out[r0:r0 + 1024, :n_per].reshape(-1).view(np.float16).astype(np.float64)
        x = vals.astype(np.float64)
        ref = (0.5 * x * (1.0 + np.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))).astype(np.float16)
        ref = np.where(x <= -10.0, np.float16(0), np.where(x >= 10.0, vals.astype(np.float16), ref))
        gb = out[r0:r0 + 1024, :n_per].reshape(-1).astype(np.int32); rb = ref.view(np.uint16).astype(np.int32)
        same_sign = (gb >> 15) == (rb >> 15)
        ulp = np.where(same_sign, np.abs(gb - rb), (gb & 0x7fff) + (rb & 0x7fff))
        # in the negative tail 1 + tanh cancels (in ggml's fp32 table as in the kernel's h - h/(exp(2u) + 1)): there the
        # bound is absolute, 5e-7, a few ulps of fp16 subnormals
        bad = (ulp > 1) & (np.abs(got - ref.astype(np.float64)) > 5e-7)
        assert not bad.any(), (int(bad.sum()), vals[bad][:4], got[bad][:4], ref[bad][:4])
        assert (ulp != 0).mean() <= 0.01, float((ulp != 0).mean())
        big = np.abs(x) >= 10.0
        assert np.array_equal(gb[big] & 0x7fff, rb[big] & 0x7fff)              # exact in the guarded ranges (up to the sign of zero)
    assert not out[:, n_per:64].any() and not out[:, 64:].any()              # gelu(0) = 0 everywhere else


@pytest.mark.parametrize("T", [1500, 100, 40, 64, 192, 33, 97, 257, 300])
@pytest.mark.parametrize("shape", ["random", "rising", "spike", "falling", "low_start"])
def test_encoder_attention_kernel_against_float64_softmax(wr, T, shape):
    """The encoder's attention kernel on score profiles that move its reference exponent: the kernel takes the maximum of a
    query's first 32 keys as the reference and moves it only when a later half-tile of 32 keys exceeds it by a margin, so
    bounded random data never leaves the fast path. `rising` climbs 30 - 90 units over the keys (many moves), `spike` puts one key
    16 - 48 units above the rest late in the sequence, `falling` makes everything after the first tile underflow, `low_start`
    starts 24 - 72 units below the rest. Lengths: the encoder's own 1500; a ragged last half-tile (100, 40, 33, 97, 300), whole
    tiles (64, 192), a padded second half of the last tile (33, 97, 257), more than one 256-query workgroup per head with query
    rows past the padded length (257, 300). Reference: softmax in float64 on the fp16 inputs; the kernel rounds p to fp16
    before the PV product: 2^-11 relative."""
    import ctypes as C
    L = wr.load_library()
    L.wmi_debug_enc_attention.restype = C.c_int
    L.wmi_debug_enc_attention.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    B, H = 2, 3
    rng = np.random.default_rng(T * 10 + len(shape))
    q = rng.standard_normal((B, H, T, 64)) * 0.35
    k = rng.standard_normal((B, H, T, 64)) * 0.7
    v = rng.standard_normal((B, H, T, 64))
    u = np.zeros(64); u[:8] = 1.0                                          # q.u = 8 a, k.u = 8 c: an additive score a*c*8
    j = np.arange(T)[None, None, :, None] / max(T - 1, 1)
    qa = 1.0 + 0.5 * (np.arange(T) % 5)[None, None, :, None]               # queries of one wave differ in how hard they are hit
    if shape != "random":
        q[..., :8] = qa * 0.5
    if shape == "rising":
        k[..., :8] = j * 7.5
    elif shape == "spike":
        k[..., :8] = 0.0; k[:, :, (2 * T) // 3, :8] = 4.0
    elif shape == "falling":
        k[..., :8] = -10.0 * (np.arange(T)[None, None, :, None] >= 64); k[:, :, :64, :8] = 0.0
    elif shape == "low_start":
        k[..., :8] = 0.0; k[:, :, :min(64, T // 2), :8] = -6.0
    q16, k16, v16 = (np.ascontiguousarray(x.astype(np.float16)) for x in (q, k, v))
    out = np.empty((B, T, H * 64), np.uint16)
    assert L.wmi_debug_enc_attention(B, H, T, q16.ctypes.data_as(C.c_void_p), k16.ctypes.data_as(C.c_void_p),
                                     v16.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    s = np.einsum("bhqd,bhkd->bhqk", q16.astype(np.float64), k16.astype(np.float64))
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhkd->bqhd", p, v16.astype(np.float64)).reshape(B, T, H * 64)
    got = out.view(np.float16).astype(np.float64)
    assert np.isfinite(got).all()
    span = float(s.max() - s.min())
    err = float(np.abs(got - ref).max())
    assert err <= 4e-3, (shape, T, err, span)                              # observed <= 2e-3 at |v| <= 4.5: twice that


@pytest.mark.parametrize("name", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"])
def test_few_row_gemm_on_blocks_equals_gemm_on_their_expansion(wr, name):
    """The few-row decoder GEMM reading a weight matrix as ggml blocks (expanded in registers) against the same kernel reading
    the load-time fp16 expansion of those blocks: identical fp32 results, bit for bit, for quantised data and for arbitrary
    block bytes (every nibble, fifth-bit and sign pattern), with K split into slices (K = 1280, 3072) and unsplit (K = 64),
    1 and 13 rows; and both equal A W^T computed in float64 from the expansion, to fp32 accumulation accuracy."""
    import ctypes as C
    from handy_amd import ggml_file as gf
    tt = getattr(gf, "T_" + name)
    L = wr.load_library()
    L.wmi_debug_quant_rows.restype = C.c_int
    L.wmi_debug_quant_rows.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(100 + tt)
    bb = gf.TTYPE_BLOCK_BYTES[tt]
    for (N, K, rows) in ((48, 64, 1), (128, 1280, 13), (64, 3072, 1), (32, 384, 16)):
        n_blocks = N * K // 32
        w = (rng.standard_normal(n_blocks * 32) * 0.05).astype(np.float32)
        raws = [gf.quantize(w, tt)]
        noise = rng.integers(0, 256, (n_blocks, bb), dtype=np.uint8)
        scale = (rng.standard_normal((n_blocks, 2)) * 0.1).astype(np.float16)
        noise[:, 0:2] = scale[:, 0:1].view(np.uint8)
        if name in ("Q4_1", "Q5_1"):
            noise[:, 2:4] = scale[:, 1:2].view(np.uint8)
        raws.append(noise.tobytes())
        a = (rng.standard_normal((rows, K)) * 0.5).astype(np.float16)
        for raw in raws:
            buf = np.frombuffer(raw, np.uint8)
            yq = np.empty((rows, N), np.float32); yf = np.empty((rows, N), np.float32)
            assert L.wmi_debug_quant_rows(tt, buf.ctypes.data_as(C.c_void_p), N, K, a.ctypes.data_as(C.c_void_p), rows,
                                          yq.ctypes.data_as(C.c_void_p), yf.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(yq.view(np.uint32) & 0x7FFFFFFF | (yq != 0) * (yq.view(np.uint32) & 0x80000000), yf.view(np.uint32) & 0x7FFFFFFF | (yf != 0) * (yf.view(np.uint32) & 0x80000000)), \
                (name, N, K, rows, float(np.abs(yq - yf).max()))           # same bits (the sign of an exact zero aside)
            wref = gf.dequantize(raw, tt, n_blocks * 32).astype(np.float16).astype(np.float64).reshape(N, K)
            ref = a.astype(np.float64) @ wref.T
            assert np.abs(yf - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (name, N, K, float(np.abs(yf - ref).max()))


@pytest.mark.parametrize("name", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"])
def test_block_expansion_is_bit_exact(wr, name):
    """The loader's device-side expansion of ggml block formats against the numpy expansion of the same bytes: identical
    fp16 bit patterns, on quantised data and on arbitrary block bytes (every nibble, high bit and sign pattern)."""
    import ctypes as C
    from handy_amd import ggml_file as gf
    tt = getattr(gf, "T_" + name)
    L = wr.load_library()
    rng = np.random.default_rng(tt)
    n_blocks = 70001                                                   # not a multiple of the kernel's 256 blocks per workgroup
    x = (rng.standard_normal(n_blocks * 32) * rng.choice([1e-3, 0.05, 3.0], n_blocks).repeat(32)).astype(np.float32)
    raws = [gf.quantize(x, tt)]
    bb = gf.TTYPE_BLOCK_BYTES[tt]
    noise = rng.integers(0, 256, (n_blocks, bb), dtype=np.uint8)
    scale = (rng.standard_normal((n_blocks, 2)) * 0.1).astype(np.float16)        # finite scales (and minimums), random quants
    noise[:, 0:2] = scale[:, 0:1].view(np.uint8)
    if name in ("Q4_1", "Q5_1"):
        noise[:, 2:4] = scale[:, 1:2].view(np.uint8)
    raws.append(noise.tobytes())
    for raw in raws:
        ref = gf.dequantize(raw, tt, n_blocks * 32).astype(np.float16)
        out = np.empty(n_blocks * 32, np.uint16)
        buf = np.frombuffer(raw, np.uint8)
        assert L.wmi_debug_expand_blocks(tt, buf.ctypes.data_as(C.c_void_p), n_blocks, out.ctypes.data_as(C.c_void_p)) == 0
        bad = np.flatnonzero(out != ref.view(np.uint16))
        assert bad.size == 0, (name, bad.size, [hex(v) for v in out[bad[:6]]], [hex(v) for v in ref.view(np.uint16)[bad[:6]]], np.frombuffer(raw, np.uint8).reshape(n_blocks, bb)[bad[0] // 32][:2], bad[:6] % 32)
    assert L.wmi_debug_expand_blocks(5, buf.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p)) == -1   # not a block type


@pytest.mark.parametrize("arch,n_mels", [("micro.en", 80), ("micro-v3", 128)])
def test_mel_parity(wr, oracle, golden, model_path, arch, n_mels):
    path = model_path(arch)
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    for seed, secs in ((0, 1.0), (1, 5.0), (2, 29.9), (3, 30.0), (4, 0.03), (5, 12.3456)):
        n = int(round(secs * 16000))
        pcm = synth_pcm(seed, n)
        ref = ost.pcm_to_mel(pcm)
        st.pcm_to_mel(pcm)
        assert st.n_len() == ost.n_len_org
        got = st.get_mel(ref.shape[1])
        err = np.abs(got - ref).max()
        assert err <= 1e-3, (seed, secs, err)          # tolerance stated by the parity contract: log-mel max abs error <= 1e-3
        key = f"mel{n_mels}_s{seed}"
        if key in golden:
            g = golden[key]
            assert np.abs(got[:, :g.shape[1] * 8:8][:, :g.shape[1]] - g).max() <= 1e-3
    # silence and a full-scale square wave (clamp edge cases)
    for pcm in (np.zeros(16000, np.float32), np.sign(np.sin(np.arange(32000) * 0.05)).astype(np.float32)):
        ref = ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
        assert np.abs(st.get_mel(ref.shape[1]) - ref).max() <= 1e-3
    st.free(); ctx.free(); ost.close(); om.close()


def test_mel_parity_dense_filter_table(wr, oracle, tmp_path):
    """A model file whose mel table has no zeros (every bin feeds every mel): the kernel reads such a table densely
    instead of from its packed on-chip copy; same contract, <= 1e-3 against the oracle."""
    from handy_amd import ggml_file as gf
    rng = np.random.default_rng(5)
    filters = (rng.random((80, 201), dtype=np.float32) * 0.01 + 1e-4).astype(np.float32)
    path = str(tmp_path / "dense_mel.bin")
    gf.write_synthetic_model(path, "micro.en", seed=0, filters=filters)
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    for seed, n in ((0, 16000), (1, 77777), (2, 480000)):
        pcm = synth_pcm(seed, n)
        ref = ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
        assert st.n_len() == ost.n_len_org
        assert np.abs(st.get_mel(ref.shape[1]) - ref).max() <= 1e-3
    st.free(); ctx.free(); ost.close(); om.close()


@pytest.mark.parametrize("arch", ["micro.en", "micro-v3", "mini"])      # "mini": d = 384, the decoder's split-K GEMMs are active
def test_encoder_decoder_parity(wr, oracle, golden, model_path, arch):
    path = model_path(arch)
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    pcm = synth_pcm(7, 20 * 16000)
    ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
    ref = ost.encode(0)
    st.encode(0)
    got = st.encoder_output()
    err = np.abs(got - ref).max()
    print(arch, "encoder max abs err", err, "ref std", ref.std())
    # observed 1.65e-3 - 1.67e-3 on every test model (values are O(1)), with the first attention kernel and with the present one
    # alike (scripts/diag_encoder_err.py). The bound leaves 30 %: a change of rounding in the encoder shows up here, not first
    # in the greedy streams downstream.
    assert err <= 2.2e-3, err
    if f"{arch}_enc" in golden:
        assert np.abs(got[::50] - golden[f"{arch}_enc"]).max() <= 8e-3       # transformers fp32 vs oracle 2.1e-3, plus the above
    kv_err = 0.0
    for layer in range(om.n_text_layer):
        k_ref, v_ref = ost.cross_kv(layer)
        k, v = st.cross_kv(layer)
        kv_err = max(kv_err, float(np.abs(k - k_ref).max()), float(np.abs(v - v_ref).max()))
    print(arch, "cross K/V max abs err", kv_err)
    assert kv_err <= 8e-3, kv_err
    toks = [int(t) for t in golden[f"{arch}_tokens" if f"{arch}_tokens" in golden else "micro.en_tokens"]]
    lref = ost.decode(toks, 0, 0, logits_all=True)
    lgot = st.decode_logits_all(toks, 0)
    lerr = np.abs(lgot - lref).max()
    print(arch, "logits max abs err", lerr, "std", lref.std())
    assert lerr <= 6e-3, lerr                                 # observed 1.9e-3 - 3.1e-3; twice that
    if f"{arch}_logits_sub" in golden:
        assert np.abs(lgot[:, ::97] - golden[f"{arch}_logits_sub"]).max() <= 1e-2
    # incremental decode through the paged cache equals the batched pass
    st.decode(toks[:4], 0)
    inc = np.stack([st.decode([toks[i]], i) for i in range(4, 8)])
    assert np.abs(inc - lgot[4:8]).max() <= 2e-3
    st.free(); ctx.free(); ost.close(); om.close()


@pytest.mark.parametrize("arch", ["mini", "mini512", "mini768"])
def test_decoder_pass_at_full_row_capacity_small_widths(wr, oracle, model_path, arch):
    """One decoder pass of 250 rows (a long prompt: nearly the 256 rows a default state holds) at tiny's, base's and small's
    widths, where the QKV and fc1 projections go through split-K with 3d- and 4d-wide partial sums: logits against the
    oracle. (The partial-sum workspace has to hold the widest of them at the state's full row count.)"""
    path = model_path(arch)
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    pcm = synth_pcm(9, 6 * 16000)
    ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
    ost.encode(0); st.encode(0)
    rng = np.random.default_rng(17)
    toks = [om.special("sot")] + [int(t) for t in rng.integers(0, 50000, size=249)]
    lref = ost.decode(toks, 0, 0, logits_all=True)
    lgot = st.decode_logits_all(toks, 0)
    err = float(np.abs(lgot - lref).max())
    print(arch, "250-row pass: logits max abs err", err, "std", float(lref.std()))
    assert err <= 1e-2, err
    # and the cache it left behind serves the next position
    nxt = st.decode([1234], 250)
    assert float(np.abs(nxt - ost.decode([1234], 250, 0)[0]).max()) <= 1e-2
    # the same prefix in passes of 1 .. 16 rows (one launch per projection with the LayerNorm in its prologue),
    # 17 .. 64 rows (split-K GEMMs with their summing consumers) and more (fused epilogues accumulating
    # slice by slice): the three paths make the same additions in the same order, so the logits are the same bits
    for n in (1, 3, 8, 16, 17, 64, 65):
        short = st.decode_logits_all(toks[:n], 0)
        assert np.array_equal(short, lgot[:n]), (n, float(np.abs(short - lgot[:n]).max()))
    st.free(); ctx.free(); ost.close(); om.close()
