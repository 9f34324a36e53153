"""CPU suite, part 1: the oracle (oracle/whisper_oracle.c) against the committed Hugging Face fixtures
and hand-built known answers. No GPU, no product code under test here."""
import ctypes as C

import numpy as np
import pytest

from handy_amd.synth import synth_pcm


@pytest.mark.parametrize("arch,n_mels", [("micro.en", 80), ("micro-v3", 128)])
def test_oracle_mel_matches_hf_fixture(oracle, golden, model_path, arch, n_mels):
    om = oracle.OracleModel(model_path(arch)); st = oracle.OracleState(om)
    for seed, secs in ((0, 1.0), (1, 5.0), (2, 29.9), (3, 30.0)):
        pcm = synth_pcm(seed, int(round(secs * 16000)))
        mel = st.pcm_to_mel(pcm)
        g = golden[f"mel{n_mels}_s{seed}"]
        assert mel.shape[1] == (len(pcm) + 480000) // 160
        assert st.n_len_org == 1 + (len(pcm) + 200 - 400) // 160
        assert np.abs(mel[:, ::8][:, :g.shape[1]] - g).max() <= 1e-3
        # frames that see no audio carry the floor value after the (max - 8) clamp
        tail = mel[:, st.n_len_org + 4:]
        assert np.all(tail == tail[0, 0]) and tail[0, 0] == pytest.approx((max(mel.max() * 4 - 4 - 8, -10.0) + 4) / 4, abs=1e-6)
    st.close(); om.close()


@pytest.mark.parametrize("arch", ["micro.en", "micro", "micro-v3"])
def test_oracle_model_matches_hf_fixture(oracle, golden, model_path, arch):
    om = oracle.OracleModel(model_path(arch)); st = oracle.OracleState(om)
    st.pcm_to_mel(synth_pcm(7, 20 * 16000))
    enc = st.encode(0)
    assert np.abs(enc[::50] - golden[f"{arch}_enc"]).max() <= 5e-3       # fp16-rounded activations vs HF fp32
    toks = [int(t) for t in golden[f"{arch}_tokens"]]
    logits = st.decode(toks, 0, 0, logits_all=True)
    assert np.abs(logits[:, ::97] - golden[f"{arch}_logits_sub"]).max() <= 5e-3
    assert (logits.argmax(-1) == golden[f"{arch}_logits_argmax"]).mean() >= 0.75    # random weights: near-ties flip
    st.decode(toks[:3], 0, 1)
    inc = np.stack([st.decode([toks[i]], i, 1)[0] for i in range(3, 8)])
    assert np.abs(inc - logits[3:8]).max() <= 1e-5                          # KV cache == full pass
    st.close(); om.close()


def test_special_tokens_and_vocab(oracle, model_path):
    exp = {"micro.en": dict(eot=50256, sot=50257, translate=50357, transcribe=50358, solm=50359, prev=50360, nosp=50361, beg=50363),
           "micro": dict(eot=50257, sot=50258, translate=50358, transcribe=50359, solm=50360, prev=50361, nosp=50362, beg=50364),
           "micro-v3": dict(eot=50257, sot=50258, translate=50359, transcribe=50360, solm=50361, prev=50362, nosp=50363, beg=50365)}
    for arch, ids in exp.items():
        om = oracle.OracleModel(model_path(arch))
        for k, v in ids.items():
            assert om.special(k) == v, (arch, k)
        assert om.special("not") == ids["beg"] - 1
        assert om.special("space") == 220 and om.token_bytes(220) == b" "
        assert om.token_bytes(ids["eot"]) == b"[_EOT_]" and om.token_bytes(ids["beg"] + 3) == b"[_TT_3]"
        if arch != "micro.en":
            assert om.token_bytes(ids["sot"] + 1) == b"[_LANG_en]"
        om.close()
    assert oracle.lib().wo_lang_id(b"en") == 0 and oracle.lib().wo_lang_id(b"german") == 2 and oracle.lib().wo_lang_id(b"yue") == 99
    assert oracle.lib().wo_lang_id(b"xx") == -1


def test_logits_rules_known_answers(oracle, model_path):
    """Hand-built logits through whisper_process_logits' restatement: one assertion per rule."""
    om = oracle.OracleModel(model_path("micro.en")); st = oracle.OracleState(om)
    V, beg, eot = om.n_vocab, om.special("beg"), om.special("eot")
    p = oracle.default_params(0)
    base = np.zeros(V, np.float32)
    base[1000] = 12.0            # one dominant text token, so the timestamp-mass rule leaves the text alone
    ninf = -np.inf
    # first position: blank and EOT suppressed, timestamps above max_initial_ts (1.0 s = 50 ticks) masked
    lg, lp, pr = st.process_logits(p, base, [], False, 3000, 0.0)
    assert lg[eot] == ninf and lg[220] == ninf and lg[beg + 50] > ninf and lg[beg + 51] == ninf
    for name in ("sot", "nosp", "solm", "translate", "transcribe", "prev", "not"):
        assert lg[om.special(name)] == ninf
    assert lg[om.special("sot") + 1] == ninf and lg[om.special("sot") + 100] == ninf      # language tokens
    # (the reference accumulates the softmax denominator sequentially in f32, so it is only good to ~1e-3)
    assert abs(float(np.exp(lp[np.isfinite(lp)].astype(np.float64)).sum()) - 1.0) < 5e-3
    assert np.isfinite(lg[1000]) and np.isfinite(lg[2000])
    # timestamp mass rule: the SUM of timestamp probabilities is compared with the single best text token,
    # so 51 timestamps at logit 9 (51 e^9 > e^12) take over and every text token is masked, without renormalising
    b2 = base.copy(); b2[beg:beg + 51] = 9.0
    lg2, lp2, pr2 = st.process_logits(p, b2, [], False, 3000, 0.0)
    assert np.all(lg2[:beg] == ninf) and pr2[:beg].sum() == 0 and 0.5 < float(pr2.sum()) < 0.9
    # flat logits: 51 timestamps at 2e-5 each also outweigh any single text token at 2e-5
    lgu, _, _ = st.process_logits(p, np.zeros(V, np.float32), [], False, 3000, 0.0)
    assert np.all(lgu[:beg] == ninf)
    # pairs: after "text, timestamp" only timestamps or EOT may follow
    b3 = base.copy(); b3[eot] = 15.0      # (EOT sits below token_beg, so it also has to out-weigh the timestamp mass)
    lg3, _, _ = st.process_logits(p, b3, [1000, beg + 10], True, 20, 0.0)
    assert np.all(lg3[:eot] == ninf) and lg3[eot] > ninf and lg3[beg + 10] > ninf and lg3[beg + 9] == ninf
    # pairs: after two timestamps no timestamp may follow
    lg4, _, _ = st.process_logits(p, base, [beg + 5, beg + 10], True, 20, 0.0)
    assert np.all(lg4[beg:] == ninf) and lg4[1000] > ninf
    # monotonic timestamps: has_ts with seek_delta 40 masks ticks below 20
    lg5, _, _ = st.process_logits(p, base, [1000], True, 40, 0.0)
    assert np.all(lg5[beg:beg + 20] == ninf) and lg5[beg + 20] > ninf
    # no_timestamps masks every timestamp
    p.no_timestamps = 1
    lg6, _, _ = st.process_logits(p, base, [1000], False, 3000, 0.0)
    assert np.all(lg6[beg:] == ninf)
    p.no_timestamps = 0
    # temperature divides the logits before the rules
    lg7, _, _ = st.process_logits(p, base, [1000], False, 3000, 0.5)
    assert lg7[1000] == pytest.approx(24.0)
    # forced length (benchmark control): EOT masked before n, only EOT at n
    p.force_n_tokens = 3
    lg8, _, _ = st.process_logits(p, base, [1000], False, 3000, 0.0)
    assert lg8[eot] == ninf
    lg9, lp9, _ = st.process_logits(p, base, [1000, 1001, 1002], False, 3000, 0.0)
    assert lg9[eot] == 0.0 and np.all(np.delete(lg9, eot) == ninf) and lp9[eot] == pytest.approx(0.0)
    st.close(); om.close()


def test_full_short_and_structure(oracle, model_path):
    om = oracle.OracleModel(model_path("micro.en", decisive=4.0)); st = oracle.OracleState(om)
    p = oracle.default_params(0); p.temperature_inc = 0.0; p.best_of = 1
    assert st.full(p, synth_pcm(1, 8000)) == []                 # under one second of audio: no segments
    segs = st.full(p, synth_pcm(2, 40 * 16000))
    assert len(segs) >= 1
    for a, b in zip(segs, segs[1:]):
        assert a["t0"] <= a["t1"] <= b["t1"]
    beg = om.special("beg")
    for s in segs:
        assert all(0 <= t < om.n_vocab for t in s["tokens"])
        assert s["text"] == b"".join(om.token_bytes(t) for t in s["tokens"] if t < om.special("eot"))
    # beam search with deterministic candidates is reproducible
    pb = oracle.default_params(1); pb.temperature_inc = 0.0; pb.beam_size = 3; pb.beam_candidates_topk = 1
    a = st.full(pb, synth_pcm(3, 6 * 16000)); b = st.full(pb, synth_pcm(3, 6 * 16000))
    assert [s["tokens"] for s in a] == [s["tokens"] for s in b]
    st.close(); om.close()


def test_quantised_block_formats(oracle, model_path):
    from handy_amd import ggml_file as gf
    rng = np.random.default_rng(0)
    x = rng.standard_normal(32 * 64).astype(np.float32) * 0.05
    tol = {gf.T_Q8_0: 1 / 127, gf.T_Q5_0: 1 / 15.5, gf.T_Q5_1: 1 / 31, gf.T_Q4_0: 1 / 7.5, gf.T_Q4_1: 1 / 15}
    for tt, rel in tol.items():
        raw = gf.quantize(x, tt)
        assert len(raw) == len(x) // 32 * gf.TTYPE_BLOCK_BYTES[tt]
        y = gf.dequantize(raw, tt, len(x))
        blk = np.abs(x.reshape(-1, 32)).max(axis=1).repeat(32)
        assert np.all(np.abs(x - y) <= blk * rel * 1.01 + 1e-6), tt
    # the oracle's loader expands the same bytes to the same values (through a whole model file)
    for ftype in (3, 8):
        path = model_path("micro.en", ftype=ftype)
        m = gf.read_model(path)
        om = oracle.OracleModel(path); st = oracle.OracleState(om)
        st.pcm_to_mel(synth_pcm(0, 16000 * 3))
        enc = st.encode(0)
        assert np.isfinite(enc).all()
        assert m.ttypes["encoder.blocks.0.attn.query.weight"] in (gf.T_Q4_1, gf.T_Q5_0)
        assert m.ttypes["encoder.conv1.bias"] == gf.T_F32 and m.ttypes["decoder.positional_embedding"] == gf.T_F32
        st.close(); om.close()
