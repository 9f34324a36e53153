"""GPU parity of the whole whisper_full path (greedy, temperature fallback, beam search, batching,
language detection) against the CPU oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from handy_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wr():
    from handy_amd import whisper_rs
    whisper_rs.load_library()
    return whisper_rs


def _oracle_params(oracle, wr_params, lang=b"en", **kw):
    c = wr_params.c
    p = oracle.default_params(c.strategy)
    p.strategy = c.strategy
    for f in ("offset_ms", "duration_ms", "max_tokens", "audio_ctx", "n_max_text_ctx", "temperature", "max_initial_ts", "length_penalty",
              "temperature_inc", "entropy_thold", "logprob_thold", "no_speech_thold"):
        setattr(p, f, getattr(c, f))
    for f in ("translate", "no_context", "no_timestamps", "single_segment", "print_special", "tdrz_enable", "detect_language", "suppress_blank", "suppress_nst"):
        setattr(p, f, int(getattr(c, f)))
    p.best_of = c.greedy.best_of
    p.beam_size = c.beam_search.beam_size
    p.language = c.language
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _run_pair(wr, oracle, path, fp, pcms, max_dec=8, **okw):
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_batch_state(len(pcms), max_dec, 4)
    if okw.get("force_n_tokens"):
        st.set_force_tokens(okw["force_n_tokens"])
    if "beam_candidates_topk" in okw:
        st.set_beam_candidates_sampled(not okw["beam_candidates_topk"])
    st.full_batch(fp, pcms)
    got = [st.batch_segments(u) for u in range(len(pcms))]
    fb = [st.batch_n_fallbacks(u) for u in range(len(pcms))]
    ref = []
    op = _oracle_params(oracle, fp, **okw)
    for pcm in pcms:
        ref.append(ost.full(op, pcm))
    st.free(); ctx.free(); ost.close(); om.close()
    return got, ref, fb


def _same(got, ref):
    if len(got) != len(ref):
        return False
    for g, r in zip(got, ref):
        if g.tokens != r["tokens"] or g.t0 != r["t0"] or g.t1 != r["t1"] or g.text.encode() != r["text"]:
            return False
    return True


@pytest.mark.parametrize("arch", ["micro.en", "micro", "micro-v3", "mini"])
def test_greedy_tokens_identical(wr, oracle, model_path, arch):
    path = model_path(arch, decisive=4.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0)
    fp.set_language("en")
    secs = (30.0, 12.5, 4.0, 47.0, 1.2, 30.0, 61.0, 8.0) if arch != "mini" else (30.0, 12.5, 4.0, 33.0)     # the oracle is slower at d = 384
    pcms = [synth_pcm(10 + i, int(16000 * s)) for i, s in enumerate(secs)]
    got, ref, _ = _run_pair(wr, oracle, path, fp, pcms)
    same = [_same(g, r) for g, r in zip(got, ref)]
    n_tok = sum(len(s["tokens"]) for r in ref for s in r)
    print(arch, "utterances identical:", same, "ref tokens", n_tok)
    assert n_tok > 50
    assert all(same)          # contract: identical greedy sequences on >= 99 % of utterances


def _decoder_state(tokens, beg):
    has_ts, seek_delta = False, 3000
    for t in tokens:
        if t > beg:
            has_ts, seek_delta = True, 2 * (t - beg)
    return has_ts, seek_delta


@pytest.mark.parametrize("arch", ["micro.en", "micro", "micro-v3"])
def test_greedy_streams_on_varied_weights(wr, oracle, model_path, arch):
    """"Varied" synthetic weights (ggml_file.synthetic_tensors) decode into long, audio-dependent streams of text tokens,
    timestamps and segment ends over several windows with carried text context -- and, unlike trained weights, leave the
    best token only a few hundredths of a logit ahead of the next, so over hundreds of steps the fp16 rounding differences
    between two implementations flip a choice now and then. The whole raw token stream of every window is recorded on both
    sides through logits_filter_callback. Where the streams stay together they must be identical; where they part, the two
    logit rows at that step must agree: these weights amplify rounding (cross-attention scores 16 times, positions 30 times
    the plain ones'), so that an encoder output that is 1.7e-3 from the oracle's moves a logit row by 0.02 - 0.17 at a logit
    spread of 2.7 on most parting steps and by more on a few (0.27 on one of the seven parting steps of the three
    architectures; which steps part, and by how much, changes with every change of the encoder's rounding: an
    attention kernel that kept a running maximum gave 0.02 - 0.17 on six, at the same encoder error of 1.65e-3 max / 2.6e-4
    rms). The bound is 0.15 on the median over an architecture's parting steps, 0.35 on each, and at most one above 0.25. And the oracle's rules applied to the ENGINE's row must pick the engine's token: the difference is then one
    of rounding, not of rules or bookkeeping. (scripts/diag_encoder_err.py prints the encoder's distance from the oracle.)"""
    import ctypes as C
    path = model_path(arch, decisive=4.0, style="varied")
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    beg, V = om.special("beg"), om.n_vocab
    n_same = n_tok = 0
    errs = []
    secs = (30.0, 12.5, 4.0, 47.0, 30.0, 61.0, 8.0)
    for i, s in enumerate(secs):
        pcm = synth_pcm(10 + i, int(16000 * s))

        def run_engine(capture=None):
            streams, row = [], {}
            def rec(tokens, logits):
                if len(tokens) == 0:
                    streams.append([])
                streams[-1][:] = [t.id for t in tokens]
                if capture == (len(streams) - 1, len(tokens)):
                    row["x"] = logits.copy()
            fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en")
            fp.set_filter_logits_callback(rec)
            st.full(fp, pcm)
            return streams, row.get("x"), [(g.t0, g.t1, g.tokens) for g in st.batch_segments(0)]

        def run_oracle(capture=None):
            streams, row = [], {}
            def rec(tokens, n, logits, _ud):
                if n == 0:
                    streams.append([])
                streams[-1][:] = [tokens[k] for k in range(n)]
                if capture == (len(streams) - 1, n):
                    row["x"] = np.ctypeslib.as_array(logits, shape=(V,)).copy()
            cb = oracle.LOGITS_FILTER(rec)
            op = oracle.default_params(0); op.temperature_inc = 0.0; op.best_of = 1; op.language = b"en"
            op.logits_filter_callback = C.cast(cb, C.c_void_p)
            segs = ost.full(op, pcm)
            return streams, row.get("x"), [(g["t0"], g["t1"], g["tokens"]) for g in segs]

        es, _, eseg = run_engine()
        os_, _, oseg = run_oracle()
        n_tok += sum(len(w) for w in os_)
        where = None
        for w in range(min(len(es), len(os_))):
            a, b = es[w], os_[w]
            j = next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), None)
            if j is None and len(a) != len(b):
                j = min(len(a), len(b))          # one side sampled a last token the callback of the other never showed
            if j is not None:
                where = (w, j); break
        if where is None and len(es) == len(os_):
            # same stream, same segments. (A window that opens with a text token takes its start time from the most probable
            # timestamp at that step, a by-product with near-ties of its own: start times are compared from the second segment on.)
            assert [(g[1], g[2]) for g in eseg] == [(g[1], g[2]) for g in oseg], (i, eseg[:2], oseg[:2])
            assert [g[0] for g in eseg[1:]] == [g[0] for g in oseg[1:]] or len(es) > 1
            n_same += 1
            continue
        assert where is not None, (i, len(es), len(os_))
        w, j = where
        if j >= min(len(es[w]), len(os_[w])):
            j -= 1                                  # the step whose outcome differed is the one before the missing call
        _, erow, _ = run_engine(capture=(w, j))
        _, orow, _ = run_oracle(capture=(w, j))
        both = np.isfinite(erow) & np.isfinite(orow)
        assert np.array_equal(np.isfinite(erow), np.isfinite(orow))
        err = float(np.abs(erow[both] - orow[both]).max())
        prefix = os_[w][:j]
        has_ts, seek_delta = _decoder_state(prefix, beg)
        picks = []
        for row in (erow, orow):
            _, lp, _ = ost.process_logits(oracle.default_params(0), row, prefix, has_ts, seek_delta, 0.0)
            picks.append(int(np.argmax(lp)))
        e_tok = es[w][j] if j < len(es[w]) else None
        o_tok = os_[w][j] if j < len(os_[w]) else None
        print(arch, "clip", i, f"streams part in window {w} at step {j}: engine {e_tok} oracle {o_tok}; rows differ by {err:.4f}; "
              f"oracle rules on the engine's row pick {picks[0]}, on its own row {picks[1]}")
        errs.append(err)
        # observed over the 21 clips of the three architectures: 0.03, 0.03, 0.06, 0.09, 0.10, 0.15 and one 0.27 (micro, clip 5, step 79 of
        # its window: 79 cached steps of fp16 K/V behind the row); the encoder output behind them is within 1.7e-3 of the oracle's
        # (test_encoder_decoder_parity bounds it at 2.2e-3)
        assert err <= 0.35, err
        if e_tok is not None and o_tok is not None:
            assert picks[0] == e_tok and picks[1] == o_tok, (picks, e_tok, o_tok)
    print(arch, "streams identical on", n_same, "of", len(secs), "clips;", n_tok, "oracle tokens")
    assert n_tok > 100
    assert sum(e > 0.25 for e in errs) <= 1, errs
    if len(errs) >= 3:
        assert float(np.median(errs)) <= 0.15, errs
    st.free(); ctx.free(); ost.close(); om.close()


def test_greedy_identical_rate_over_many_utterances(wr, oracle, model_path):
    """The parity contract's rate -- greedy token sequences identical on >= 99 % of utterances -- over 160 seeded clips of
    1.5 to 14 s in one ragged batch (timestamps on, text context carried), not just a handful."""
    path = model_path("micro.en", decisive=4.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0)
    fp.set_language("en")
    rng = np.random.default_rng(123)
    secs = rng.uniform(1.5, 14.0, size=160)
    pcms = [synth_pcm(1000 + i, int(16000 * s)) for i, s in enumerate(secs)]
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_batch_state(len(pcms), 1, 16)
    st.full_batch(fp, pcms)
    op = _oracle_params(oracle, fp)
    same, n_tok = 0, 0
    for u, pcm in enumerate(pcms):
        ref = ost.full(op, pcm)
        same += _same(st.batch_segments(u), ref)
        n_tok += sum(len(s["tokens"]) for s in ref)
    st.free(); ctx.free(); ost.close(); om.close()
    print("identical utterances:", same, "of", len(pcms), "| reference tokens:", n_tok)
    assert n_tok > 300
    assert same >= 0.99 * len(pcms), (same, len(pcms))


def test_greedy_default_params_with_fallback(wr, oracle, model_path):
    """Default whisper_full parameters. Random weights give low average log-probabilities, so every window walks the
    whole temperature ladder and the T>0 path (host-side draw with the decoder's mt19937 over the device-computed
    probabilities) runs. A draw from a near-uniform 51864-way distribution flips on last-digit differences in the
    probabilities, so token identity with an independent implementation is not defined there; what is checked is the
    control flow (same number of fallbacks as the oracle) and that the result is well formed."""
    path = model_path("micro.en", decisive=1.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(2))
    pcms = [synth_pcm(40 + i, int(16000 * s)) for i, s in enumerate((9.0, 30.0, 3.0))]
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ref_fb = []
    for pcm in pcms:
        ost.full(_oracle_params(oracle, fp), pcm)
        ref_fb.append(ost.n_fallbacks)
    ctx = wr.WhisperContext(path); st = ctx.create_batch_state(len(pcms), 2, 2)
    st.full_batch(fp, pcms)
    fb = [st.batch_n_fallbacks(u) for u in range(len(pcms))]
    print("fallbacks", fb, "oracle", ref_fb)
    assert fb == ref_fb and sum(fb) > 0
    for u in range(len(pcms)):
        segs = st.batch_segments(u)
        assert all(0 <= t < om.n_vocab for s in segs for t in s.tokens)
        assert all(s.t1 >= s.t0 for s in segs)
    st.free(); ctx.free(); ost.close(); om.close()


@pytest.mark.parametrize("topk", [1, 0])
def test_beam_search_matches_oracle(wr, oracle, model_path, topk):
    path = model_path("micro.en", decisive=4.0, style="varied")
    fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(3, -1.0))
    fp.set_temperature_inc(0.0)
    pcms = [synth_pcm(60 + i, int(16000 * s)) for i, s in enumerate((10.0, 30.0, 5.5))]
    got, ref, _ = _run_pair(wr, oracle, path, fp, pcms, beam_candidates_topk=topk)
    same = [_same(g, r) for g, r in zip(got, ref)]
    print("beam topk=%d identical" % topk, same)
    assert all(same)
    assert sum(len(s["tokens"]) for r in ref for s in r) >= 6


def test_forced_length_and_no_timestamps(wr, oracle, model_path):
    path = model_path("micro-v3", decisive=2.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0); fp.set_no_timestamps(True); fp.set_language("de")
    pcms = [synth_pcm(80 + i, 30 * 16000) for i in range(4)]
    got, ref, _ = _run_pair(wr, oracle, path, fp, pcms, force_n_tokens=48)
    for g, r in zip(got, ref):
        assert _same(g, r)
        assert sum(len(s.tokens) for s in g) == 49          # 48 tokens + the forced end-of-text


def test_batch_equals_single_and_ragged_inputs(wr, model_path):
    path = model_path("micro", decisive=4.0)
    ctx = wr.WhisperContext(path)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("fr")
    pcms = [synth_pcm(100 + i, n) for i, n in enumerate((480000, 16000, 15999, 400, 1, 250000, 720001))]
    bst = ctx.create_batch_state(len(pcms), 1, 3)
    bst.full_batch(fp, pcms)
    batch = [bst.batch_tokens(u) for u in range(len(pcms))]
    single = []
    st = ctx.create_state()
    for pcm in pcms:
        st.full(fp, pcm)
        single.append(st.batch_tokens(0))
    assert batch == single
    assert batch[3] == [] and batch[4] == []          # under one second of audio: no output, no error
    assert len(batch[0]) > 0
    with pytest.raises(wr.WhisperError):
        st.full(fp, np.zeros(0, np.float32))
    bst.free(); st.free(); ctx.free()


VARIANTS = {
    "audio_ctx": lambda fp: fp.set_audio_ctx(768),
    "audio_ctx_600": lambda fp: fp.set_audio_ctx(600),         # padded to 640 keys: the last 256-query block of the attention reaches past the padding
    "translate": lambda fp: fp.set_translate(True),
    "single_segment": lambda fp: fp.set_single_segment(True),
    "max_tokens": lambda fp: fp.set_max_tokens(20),
    "offset_duration": lambda fp: (fp.set_offset_ms(5000), fp.set_duration_ms(12000)),
    "prompt_tokens": lambda fp: fp.set_tokens([1001, 2002, 3003, 404, 505]),
    "print_special": lambda fp: fp.set_print_special(True),
    "no_blank_rule_nst_rule": lambda fp: (fp.set_suppress_blank(False), fp.set_suppress_nst(True)),
    "max_initial_ts": lambda fp: fp.set_max_initial_ts(0.5),
    "tdrz": lambda fp: fp.set_tdrz_enable(True),
    "no_timestamps": lambda fp: fp.set_no_timestamps(True),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_full_params_variants_match_oracle(wr, oracle, model_path, variant):
    """One whisper_full parameter changed at a time (the ones transcribe-rs / whisper-rs callers can reach), greedy,
    multilingual model, against the oracle: identical tokens, timestamps and text."""
    path = model_path("micro", decisive=4.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0)
    fp.set_language("de")
    VARIANTS[variant](fp)
    pcms = [synth_pcm(300 + i, int(16000 * s)) for i, s in enumerate((30.0, 17.0, 41.0))]
    okw = {}
    if variant == "prompt_tokens":
        arr = (C.c_int32 * 5)(1001, 2002, 3003, 404, 505)
        okw = {"prompt_tokens": C.cast(arr, C.POINTER(C.c_int32)), "prompt_n_tokens": 5, "_keep": arr}
    keep = okw.pop("_keep", None)
    got, ref, _ = _run_pair(wr, oracle, path, fp, pcms, **okw)
    same = [_same(g, r) for g, r in zip(got, ref)]
    print(variant, same, [sum(len(s["tokens"]) for s in r) for r in ref])
    assert all(same)
    assert sum(len(s["tokens"]) for r in ref for s in r) > 10
    del keep


def test_initial_prompt_text_goes_through_the_tokenizer(wr, model_path):
    """`initial_prompt` (text) and `prompt_tokens` (ids) are the same request once the text is tokenised: whisper_tokenize through
    the C ABI gives the ids of the word-pattern / longest-entry rule (restated with Python's `re`; the rule itself is held to
    that restatement on many texts, under the sanitizers, in tests/test_cpu_sanitizers.py), and a transcription prompted with
    the text equals one prompted with those ids and differs from an unprompted one."""
    import re
    from handy_amd import ggml_file
    path = model_path("micro", decisive=4.0)
    vocab = ggml_file.read_model(path).vocab
    ids = {t: i for i, t in enumerate(vocab)}
    pat = re.compile(rb"'s|'t|'re|'ve|'m|'ll|'d| ?[A-Za-z]+| ?[0-9]+| ?[^\sA-Za-z0-9]+|\s+(?!\S)|\s+")
    words = [t.decode() for t in vocab[300:330]]
    text = "".join(w if w.startswith(" ") else " " + w for w in words[:12]) + "'s 42, ok?  "
    want = []
    for w in pat.findall(text.encode()):
        i = 0
        while i < len(w):
            for j in range(len(w), i, -1):
                if w[i:j] in ids:
                    want.append(ids[w[i:j]]); i = j
                    break
            else:
                i += 1
    ctx = wr.WhisperContext(path)
    got = ctx.tokenize(text)
    assert got == want and len(got) >= 12, (got, want)
    pcm = synth_pcm(77, 16000 * 20)

    def run(setup):
        st = ctx.create_batch_state(1, 1, 1)
        fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en")
        keep = setup(fp)
        st.full_batch(fp, [pcm])
        out = [(g.tokens, g.t0, g.t1) for g in st.batch_segments(0)]
        st.free()
        del keep
        return out

    by_text = run(lambda fp: fp.set_initial_prompt(text))
    by_ids = run(lambda fp: fp.set_tokens(got))
    plain = run(lambda fp: None)
    ctx.free()
    assert by_text == by_ids and sum(len(t) for t, _, _ in by_text) > 5
    assert by_text != plain


def test_low_level_entry_points(wr, model_path):
    """The rest of the whisper.h surface whisper-rs binds, each through the C ABI: a context from a memory buffer transcribes as
    the one from the file does; a log-mel handed in with whisper_set_mel_with_state encodes to the same bits as the one the
    engine computed itself (and whisper_n_len_from_state counts its frames); whisper_full_default_params_by_ref returns what
    the by-value call returns; whisper_print_system_info names the device path; whisper_print_timings / whisper_reset_timings
    go through the log callback; speaker_turn_next reads false without tinydiarize tokens."""
    path = model_path("micro", decisive=4.0)
    L = wr.load_library()
    pcm = synth_pcm(21, 16000 * 12)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en")

    def transcript(ctx):
        st = ctx.create_batch_state(1, 1, 1)
        st.full_batch(fp, [pcm])
        out = [(g.tokens, g.t0, g.t1) for g in st.batch_segments(0)]
        st.free()
        return out

    ctx = wr.WhisperContext(path)
    ctx_b = wr.WhisperContext.from_buffer(open(path, "rb").read())
    a, b = transcript(ctx), transcript(ctx_b)
    assert a == b and sum(len(t) for t, _, _ in a) > 3
    ctx_b.free()
    with pytest.raises(wr.WhisperError):
        wr.WhisperContext.from_buffer(open(path, "rb").read()[:5000])            # a truncated image is refused, not read past

    s1 = ctx.create_state(); s2 = ctx.create_state()
    s1.pcm_to_mel(pcm)
    n = s1.n_len()
    assert n == 1 + (pcm.size + 200 - 400) // 160                                 # whisper.cpp reports the frames of the audio itself (mel.n_len_org)
    mel = s1.get_mel(3000)
    s1.encode(0)
    e1 = s1.encoder_output()
    s2.set_mel(mel)
    assert s2.n_len() == 3000
    s2.encode(0)
    e2 = s2.encoder_output()
    assert np.array_equal(e1, e2) and np.abs(e1).max() > 0.5
    with pytest.raises(wr.WhisperError):
        s2.set_mel(np.zeros((ctx.n_mels() + 1, 100), np.float32))                 # wrong number of mel bands
    # speaker turns: no tinydiarize token in this vocabulary's output
    s1.full(fp, pcm)
    assert s1.full_n_segments() >= 1
    assert not any(L.whisper_full_get_segment_speaker_turn_next_from_state(s1.h, i) for i in range(s1.full_n_segments()))
    s1.free(); s2.free()

    for strategy in (0, 1):
        pr = L.whisper_full_default_params_by_ref(strategy)
        by_val = L.whisper_full_default_params(strategy)
        assert pr and bytes(C.string_at(C.addressof(pr.contents), C.sizeof(wr.whisper_full_params))) == \
            bytes(C.string_at(C.addressof(by_val), C.sizeof(wr.whisper_full_params)))
        L.whisper_free_params(pr)
    info = L.whisper_print_system_info()
    assert info and (b"HIP" in info or b"gfx" in info or b"MI3" in info), info

    lines = []
    cb_t = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
    cb = cb_t(lambda level, text, user: lines.append(text.decode(errors="replace")))
    L.whisper_log_set.argtypes = [cb_t, C.c_void_p]; L.whisper_log_set.restype = None
    L.whisper_log_set(cb, None)
    try:
        L.whisper_print_timings(ctx.h)
        L.whisper_reset_timings(ctx.h)
    finally:
        L.whisper_log_set(C.cast(None, cb_t), None)
    assert any("ms" in t for t in lines), lines
    ctx.free()


def test_context_level_entry_points(wr, model_path):
    """whisper.h's calls that take the context and use its built-in state (whisper_init_from_file_with_params, whisper_full,
    whisper_full_parallel, the accessors without `_from_state`, whisper_pcm_to_mel / whisper_encode / whisper_decode /
    whisper_get_logits / whisper_lang_auto_detect, whisper_token_count, the model getters): the same results as the
    state-taking calls the other tests use."""
    path = model_path("micro", decisive=4.0)
    L = wr.load_library()
    vp, ci = C.c_void_p, C.c_int
    L.whisper_init_from_file_with_params.restype = vp; L.whisper_init_from_file_with_params.argtypes = [C.c_char_p, wr.whisper_context_params]
    L.whisper_context_default_params.restype = wr.whisper_context_params; L.whisper_context_default_params.argtypes = []
    L.whisper_full.restype = ci; L.whisper_full.argtypes = [vp, wr.whisper_full_params, C.POINTER(C.c_float), ci]
    L.whisper_full_parallel.restype = ci; L.whisper_full_parallel.argtypes = [vp, wr.whisper_full_params, C.POINTER(C.c_float), ci, ci]
    for n, rt in (("whisper_full_n_segments", ci), ("whisper_full_lang_id", ci), ("whisper_n_len", ci), ("whisper_model_n_vocab", ci),
                  ("whisper_model_n_audio_ctx", ci), ("whisper_model_n_text_ctx", ci), ("whisper_model_n_text_state", ci),
                  ("whisper_model_n_text_head", ci), ("whisper_model_type", ci)):
        getattr(L, n).restype = rt; getattr(L, n).argtypes = [vp]
    for n, rt in (("whisper_full_get_segment_t0", C.c_int64), ("whisper_full_get_segment_t1", C.c_int64), ("whisper_full_get_segment_text", C.c_char_p),
                  ("whisper_full_n_tokens", ci), ("whisper_full_get_segment_speaker_turn_next", C.c_bool)):
        getattr(L, n).restype = rt; getattr(L, n).argtypes = [vp, ci]
    L.whisper_full_get_token_id.restype = ci; L.whisper_full_get_token_id.argtypes = [vp, ci, ci]
    L.whisper_full_get_token_p.restype = C.c_float; L.whisper_full_get_token_p.argtypes = [vp, ci, ci]
    L.whisper_full_get_token_text.restype = C.c_char_p; L.whisper_full_get_token_text.argtypes = [vp, ci, ci]
    L.whisper_full_get_token_data.restype = wr.whisper_token_data; L.whisper_full_get_token_data.argtypes = [vp, ci, ci]
    L.whisper_pcm_to_mel.restype = ci; L.whisper_pcm_to_mel.argtypes = [vp, C.POINTER(C.c_float), ci, ci]
    L.whisper_encode.restype = ci; L.whisper_encode.argtypes = [vp, ci, ci]
    L.whisper_decode.restype = ci; L.whisper_decode.argtypes = [vp, C.POINTER(C.c_int32), ci, ci, ci]
    L.whisper_get_logits.restype = C.POINTER(C.c_float); L.whisper_get_logits.argtypes = [vp]
    L.whisper_lang_auto_detect.restype = ci; L.whisper_lang_auto_detect.argtypes = [vp, ci, ci, C.POINTER(C.c_float)]
    L.whisper_token_count.restype = ci; L.whisper_token_count.argtypes = [vp, C.c_char_p]
    L.whisper_lang_max_id.restype = ci; L.whisper_lang_max_id.argtypes = []
    L.whisper_lang_str_full.restype = C.c_char_p; L.whisper_lang_str_full.argtypes = [ci]

    pcm = np.ascontiguousarray(synth_pcm(31, 16000 * 14), np.float32)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("auto")
    # the reference result: context without a state + a state of its own
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    st.full(fp, pcm)
    want = [(st.full_get_segment_t0(i), st.full_get_segment_t1(i), [st.full_get_token_id(i, j) for j in range(st.full_n_tokens(i))])
            for i in range(st.full_n_segments())]
    want_lang = st.full_lang_id_from_state()
    st.pcm_to_mel(pcm); st.encode(0)
    sot = L.whisper_token_sot(ctx.h)
    want_logits = st.decode([sot], 0)
    want_vocab = ctx.n_vocab()
    n_tok = len(ctx.tokenize(" kato 12"))
    st.free(); ctx.free()

    h = L.whisper_init_from_file_with_params(path.encode(), L.whisper_context_default_params())
    assert h
    try:
        pp = pcm.ctypes.data_as(C.POINTER(C.c_float))
        for call in ("full", "parallel"):
            rc = L.whisper_full(h, fp.c, pp, pcm.size) if call == "full" else L.whisper_full_parallel(h, fp.c, pp, pcm.size, 1)
            assert rc == 0
            got = [(L.whisper_full_get_segment_t0(h, i), L.whisper_full_get_segment_t1(h, i),
                    [L.whisper_full_get_token_id(h, i, j) for j in range(L.whisper_full_n_tokens(h, i))]) for i in range(L.whisper_full_n_segments(h))]
            assert got == want and len(got) >= 1, (call, got, want)
            assert L.whisper_full_lang_id(h) == want_lang
            d = L.whisper_full_get_token_data(h, 0, 0)
            assert d.id == want[0][2][0] and abs(d.p - L.whisper_full_get_token_p(h, 0, 0)) < 1e-7 and 0.0 < d.p <= 1.0
            assert L.whisper_full_get_token_text(h, 0, 0) is not None and L.whisper_full_get_segment_text(h, 0) is not None
            assert L.whisper_full_get_segment_speaker_turn_next(h, 0) is False
        assert L.whisper_n_len(h) == 1 + (pcm.size + 200 - 400) // 160
        assert L.whisper_pcm_to_mel(h, pp, pcm.size, 1) == 0 and L.whisper_encode(h, 0, 1) == 0
        toks = (C.c_int32 * 1)(sot)
        assert L.whisper_decode(h, toks, 1, 0, 1) == 0
        logits = np.ctypeslib.as_array(L.whisper_get_logits(h), shape=(want_vocab,)).copy()
        assert np.array_equal(logits, want_logits)
        probs = (C.c_float * (L.whisper_lang_max_id() + 1))()
        assert L.whisper_lang_auto_detect(h, 0, 1, probs) == want_lang
        assert abs(sum(probs) - 1.0) < 1e-3
        assert L.whisper_token_count(h, b" kato 12") == n_tok > 0
        assert (L.whisper_model_n_vocab(h), L.whisper_model_n_audio_ctx(h), L.whisper_model_n_text_ctx(h)) == (want_vocab, 1500, 448)
        assert L.whisper_model_n_text_state(h) > 0 and L.whisper_model_n_text_head(h) > 0 and L.whisper_model_type(h) >= 0
        assert L.whisper_lang_str_full(0) == b"english"
    finally:
        L.whisper_free(h)


def test_batching_front_equals_one_call_at_a_time(wr, oracle, model_path):
    """handy_amd.manager.TranscriptionManager: concurrent transcribe() calls are served by shared GPU batches and
    every caller still gets exactly what a lone call returns (beam search, transcribe-rs defaults) -- which is the text the
    oracle's whisper_full produces for that clip with the same parameters."""
    import threading
    from handy_amd.manager import TranscriptionManager
    path = model_path("micro", decisive=4.0)
    pcms = [synth_pcm(400 + i, n) for i, n in enumerate((480000, 90000, 250000, 16000 * 7, 480000, 300000))]
    lone = TranscriptionManager(max_batch=1, collect_ms=0.0, selected_language="en")
    lone.load_model(path)
    expect = [lone.transcribe(p) for p in pcms]
    lone.unload_model()
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    fp = wr.WhisperEngine()._full_params(wr.WhisperInferenceParams(language="en"))      # what the manager hands to the engine
    op = _oracle_params(oracle, fp, beam_candidates_topk=1)
    ref = [b"".join(s["text"] for s in ost.full(op, p)).decode("utf-8", errors="replace").strip() for p in pcms]
    ost.close(); om.close()
    assert expect == ref
    tm = TranscriptionManager(max_batch=8, collect_ms=50.0, selected_language="en")
    tm.load_model(path)
    got = [None] * len(pcms)
    def call(i): got[i] = tm.transcribe(pcms[i])
    th = [threading.Thread(target=call, args=(i,)) for i in range(len(pcms))]
    for t in th: t.start()
    for t in th: t.join()
    assert got == expect and any(len(t) > 0 for t in expect)
    assert tm.batches_run < len(pcms)
    tm.unload_model()


@pytest.mark.parametrize("arch,lang", [("micro.en", "en"), ("micro", "auto")])
def test_window_memo_on_a_growing_recording(wr, oracle, model_path, arch, lang):
    """Handy's live coaching transcribes the whole recording so far every few seconds (reference live_coaching.rs:84-97).
    With the window memo on, the interior 30 s windows of the earlier calls are taken over instead of being encoded and
    decoded again, and every call returns exactly what it returns with the memo off -- greedy with temperature fallback
    and beam search, text context carried between windows."""
    path = model_path(arch)
    ctx = wr.WhisperContext(path)
    full = synth_pcm(41, 100 * 16000)
    full[: 8 * 16000] *= 1.0
    full[8 * 16000:] *= 0.4                    # the loudest stretch is early: the utterance-wide log-mel maximum stops moving
    lengths = [40, 71, 74, 77, 100]             # seconds recorded at each peek
    for strategy in (wr.SamplingStrategy.Greedy(2), wr.SamplingStrategy.BeamSearch(3, -1.0)):
        fp = wr.FullParams(strategy)
        fp.set_language(lang)                   # "auto": the first window is encoded for language detection on every call
        plain = ctx.create_batch_state(1, 3, 1)
        memo = ctx.create_batch_state(1, 3, 1)
        memo.set_window_cache(16)
        for secs in lengths:
            pcm = full[: secs * 16000]
            plain.full_batch(fp, [pcm]); memo.full_batch(fp, [pcm])
            a, b = plain.batch_segments(0), memo.batch_segments(0)
            assert len(a) == len(b) and len(a) > 0
            for x, y in zip(a, b):
                assert (x.t0, x.t1, x.text, x.tokens) == (y.t0, y.t1, y.text, y.tokens), secs
            assert plain.batch_n_fallbacks(0) == memo.batch_n_fallbacks(0)
        stats = memo.window_cache_stats()
        assert stats["stores"] >= 2 and stats["hits"] >= (4 if lang == "en" else 2), stats       # windows 0 and 1 are interior from the second peek on
        assert plain.window_cache_stats() == {"hits": 0, "stores": 0}
        # a different recording never hits
        other = synth_pcm(42, 71 * 16000)
        h0 = memo.window_cache_stats()["hits"]
        memo.full_batch(fp, [other])
        assert memo.window_cache_stats()["hits"] == h0
        memo.set_window_cache(0)
        memo.full_batch(fp, [full[: 77 * 16000]])
        assert memo.window_cache_stats()["hits"] == h0
        plain.free(); memo.free()
    ctx.free()
    # ... and what the memo returns is what the oracle's whisper_full returns for the recording: greedy without draws on the
    # weights with decisive margins (at temperature > 0 a draw turns on differences in the probabilities far below the
    # parity tolerance, so the comparison above is between the engine's own two paths)
    path4 = model_path(arch, decisive=4.0)
    ctx = wr.WhisperContext(path4)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language(lang)
    memo = ctx.create_batch_state(1, 1, 1)
    memo.set_window_cache(16)
    for secs in lengths:
        pcm = full[: secs * 16000]
        memo.full_batch(fp, [pcm])
        if secs >= 77:
            om = oracle.OracleModel(path4); ost = oracle.OracleState(om)
            ref = ost.full(_oracle_params(oracle, fp, lang=None, beam_candidates_topk=1), pcm)
            b = memo.batch_segments(0)
            assert _same(b, ref), (secs, [x.tokens[:6] for x in b[:3]], [r["tokens"][:6] for r in ref[:3]])
            ost.close(); om.close()
    assert memo.window_cache_stats()["hits"] >= 2
    memo.free(); ctx.free()


@pytest.mark.parametrize("arch", ["mini", "mini768"])
def test_decoder_row_lanes_do_not_change_results(wr, model_path, arch):
    """A decoder pass of 32 rows or more runs as two halves of its rows on two HIP streams. No kernel's arithmetic depends on the
    rows that share its launch, so one lane and two lanes return the same tokens, times and log-probabilities to the bit --
    greedy (one row per utterance) and beam search (three rows per utterance, cut between utterances), ragged clips."""
    ctx = wr.WhisperContext(model_path(arch, decisive=4.0, style="varied"))
    pcms = [synth_pcm(300 + i, int(16000 * (3.0 + (i * 7) % 27))) for i in range(40)]
    for strategy in (wr.SamplingStrategy.Greedy(1), wr.SamplingStrategy.BeamSearch(3, -1.0)):
        fp = wr.FullParams(strategy); fp.set_temperature_inc(0.0); fp.set_language("en")
        out = {}
        for lanes in (1, 2):
            st = ctx.create_batch_state(len(pcms), 3, 16)
            st.set_decoder_lanes(lanes)
            st.full_batch(fp, pcms)
            out[lanes] = [[(g.t0, g.t1, tuple(g.tokens), tuple(np.float32(p).tobytes() for p in g.plog)) for g in st.batch_segments(u)] for u in range(len(pcms))]
            st.free()
        assert out[1] == out[2]
        assert sum(len(s[2]) for u in out[1] for s in u) > 200
    ctx.free()


def test_two_states_on_two_threads(wr, model_path):
    """Handy calls the engine from whatever thread has work (tokio blocking tasks, the live-coaching thread; reference
    transcription.rs:383, live_coaching.rs:70, discord_conversation.rs:475): two states of one context running at the
    same time from two threads return what they return one after the other."""
    import threading
    ctx = wr.WhisperContext(model_path("mini", decisive=4.0))
    fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(3, -1.0)); fp.set_language("en")
    sets = [[synth_pcm(60 + i, int(16000 * s)) for i, s in enumerate((9.0, 31.0, 4.5))],
            [synth_pcm(70 + i, int(16000 * s)) for i, s in enumerate((20.0, 2.5, 12.0, 6.0))]]
    states = [ctx.create_batch_state(4, 3, 2) for _ in sets]
    def run(k):
        states[k].full_batch(fp, sets[k])
        return [[(s.t0, s.t1, s.text, tuple(s.tokens)) for s in states[k].batch_segments(u)] for u in range(len(sets[k]))]
    serial = [run(0), run(1)]
    for _ in range(3):
        got = [None, None]; errs = []
        def work(k):
            try: got[k] = run(k)
            except Exception as e: errs.append(e)
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errs, errs
        assert got == serial
    for st in states: st.free()
    ctx.free()


def test_pipeline_groups_do_not_change_results(wr, model_path):
    """Decoding index ranges of the batch one after the other while the encoder stream still works on the later ones
    (wmi_set_pipeline_groups) changes which sequences share a decoder pass, never their tokens."""
    path = model_path("mini", decisive=4.0)
    ctx = wr.WhisperContext(path)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en")
    pcms = [synth_pcm(200 + i, n) for i, n in enumerate((480000, 200000, 330000, 480000, 64000, 700000, 480000, 120000, 480000, 250000))]
    runs = []
    for groups in (1, 3, 10):
        st = ctx.create_batch_state(len(pcms), 1, 2)
        st.set_pipeline_groups(groups)
        st.full_batch(fp, pcms)
        runs.append([st.batch_tokens(u) for u in range(len(pcms))])
        st.free()
    assert runs[0] == runs[1] == runs[2]
    assert sum(len(t) for t in runs[0]) > 100
    ctx.free()


def test_language_detection_and_engine_mirror(wr, oracle, model_path):
    path = model_path("micro", decisive=4.0)
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    pcm = synth_pcm(123, 20 * 16000)
    op = oracle.default_params(0); op.language = b"auto"; op.temperature_inc = 0.0; op.best_of = 1
    ref = ost.full(op, pcm)
    ref_lang = ost.L.wo_full_lang_id(ost.h)
    eng = wr.WhisperEngine()
    eng.load_model(path)
    lid, probs = eng.state.lang_detect(0) if (eng.state.pcm_to_mel(pcm) is None) else (None, None)
    assert lid == ref_lang and abs(float(probs.sum()) - 1.0) < 1e-3
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("auto")
    eng.state.full(fp, pcm)
    assert eng.state.full_lang_id_from_state() == ref_lang
    assert eng.state.batch_tokens(0) == [t for s in ref for t in s["tokens"]]
    res = eng.transcribe_samples(pcm, wr.WhisperInferenceParams(language="en", beam_size=1))
    assert isinstance(res.text, str) and res.segments is not None
    eng.unload_model()
    ost.close(); om.close()


def test_quantised_files_load_and_match(wr, oracle, model_path):
    """Handy ships Q4_1 and Q5_0 files (model.rs:97,135): every block format the loader accepts."""
    pcm = synth_pcm(7, 10 * 16000)
    for ftype in (3, 8, 7, 2, 9, 0):
        path = model_path("micro.en", ftype=ftype)
        om = oracle.OracleModel(path); ost = oracle.OracleState(om)
        ctx = wr.WhisperContext(path); st = ctx.create_state()
        ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
        ref = ost.encode(0); st.encode(0)
        qerr = float(np.abs(st.encoder_output() - ref).max())
        print("ftype", ftype, "encoder max abs err", qerr)
        assert qerr <= 5e-3, (ftype, qerr)          # observed 1.3e-3 - 2.5e-3; twice that
        st.free(); ctx.free(); ost.close(); om.close()


@pytest.mark.parametrize("arch", ["micro.en", "mini", "mini768"])
def test_quantised_resident_decode_is_bit_identical(wr, oracle, model_path, arch):
    """Block-quantised files keep the decoder's matrices resident as blocks; decode passes of at most 16 rows expand them in
    registers instead of reading the fp16 copies. The expansion is the load-time one value for value, so logits and
    transcripts are the same bits either way -- every block format, widths with and without split K, one row and a
    short prompt -- and both agree with the oracle (which expands the file's blocks on the host)."""
    pcm = synth_pcm(8, 9 * 16000)
    rng = np.random.default_rng(5)
    wr.load_library().wmi_set_quant_resident(1)           # (off by default: see include/wmi.h)
    for ftype in (3, 8, 7, 2, 9):                         # Q4_1, Q5_0, Q8_0, Q4_0, Q5_1
        path = model_path(arch, ftype=ftype, decisive=4.0)
        om = oracle.OracleModel(path); ost = oracle.OracleState(om)
        ctx = wr.WhisperContext(path); st = ctx.create_state()
        ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
        ost.encode(0); st.encode(0)
        toks = [om.special("sot")] + [int(t) for t in rng.integers(0, 50000, size=11)]
        lref = ost.decode(toks, 0, 0, logits_all=True)
        got = {}
        for on in (True, False):
            st.set_quant_decode(on)
            a = st.decode_logits_all(toks, 0)              # 12 rows in one pass
            b = np.concatenate([st.decode_logits_all(toks[:5], 0)] + [st.decode([t], 5 + i)[None] for i, t in enumerate(toks[5:])])   # 5 rows, then one at a time
            assert np.array_equal(a[:5], b[:5]) and np.array_equal(a[5:], b[5:, : a.shape[1]]), (arch, ftype, on)
            got[on] = a
        assert np.array_equal(got[True], got[False]), (arch, ftype, float(np.abs(got[True] - got[False]).max()))
        err, scale = float(np.abs(got[True] - lref).max()), float(np.abs(lref).max())
        print(arch, "ftype", ftype, "logits max abs err vs oracle", err, "max |ref|", scale)
        assert err <= 1.5e-3 * max(1.0, scale), (arch, ftype, err, scale)      # observed 1.2e-2 at max |ref| 26 (decisive weights): 5e-4 of the range; three times that
        st.free(); ost.close(); om.close()
        # and whisper_full on one clip: same tokens and times with the blocks as with the fp16 copies
        fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(3, -1.0)); fp.set_language("en")
        out = {}
        for on in (True, False):
            bs = ctx.create_batch_state(1, 3, 1)
            bs.set_quant_decode(on)
            bs.full_batch(fp, [pcm])
            out[on] = [(g.t0, g.t1, tuple(g.tokens)) for g in bs.batch_segments(0)]
            bs.free()
        assert out[True] == out[False] and len(out[True]) > 0, (arch, ftype)
        ctx.free()
    wr.load_library().wmi_set_quant_resident(0)


def test_bad_model_files_fail_cleanly(wr, model_path, tmp_path):
    good = open(model_path("micro.en"), "rb").read()
    for name, data in (("trunc.bin", good[: len(good) // 2]), ("magic.bin", b"\0" * 4096), ("hdr.bin", good[:40])):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(wr.WhisperError):
            wr.WhisperContext(str(p))
    with pytest.raises(wr.WhisperError):
        wr.WhisperContext(str(tmp_path / "missing.bin"))


def test_best_of_decoders_draw_different_sequences(wr, model_path):
    """Temperature 0.8 without fallback: decoder j draws with its own generator, so the best of four sequences is, on some
    clip at least, not the sequence the lone decoder of a best_of = 1 run draws (with one seed for all of them the four
    would be copies of that one and best_of would buy nothing)."""
    path = model_path("micro.en", decisive=1.0)
    ctx = wr.WhisperContext(path)
    pcms = [synth_pcm(500 + i, 6 * 16000) for i in range(6)]
    out = {}
    for k in (1, 4):
        fp = wr.FullParams(wr.SamplingStrategy.Greedy(k))
        fp.set_temperature(0.8); fp.set_temperature_inc(0.0); fp.set_language("en")
        st = ctx.create_batch_state(len(pcms), 4, 4)
        st.full_batch(fp, pcms)
        out[k] = [st.batch_tokens(u) for u in range(len(pcms))]
        st.full_batch(fp, pcms)
        assert [st.batch_tokens(u) for u in range(len(pcms))] == out[k]          # the draws repeat run to run
        st.free()
    ctx.free()
    assert all(len(t) > 0 for t in out[1])
    assert sum(a != b for a, b in zip(out[1], out[4])) >= 1, "best_of = 4 returned the best_of = 1 sequences on every clip"


def test_text_context_carries_over_between_calls(wr, oracle, model_path):
    """no_context = false: the second whisper_full call of a state starts from the text the first one produced (the prompt
    tokens of the call go in front of it), as in the oracle; no_context = true forgets it."""
    path = model_path("micro", decisive=4.0, style="varied")
    om = oracle.OracleModel(path); ost = oracle.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    a, b = synth_pcm(600, 12 * 16000), synth_pcm(601, 9 * 16000)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en"); fp.set_no_context(False)
    op = _oracle_params(oracle, fp)
    outs = []
    for pcm in (a, b, b):
        st.full(fp, pcm)
        got = st.batch_segments(0)
        ref = ost.full(op, pcm)
        assert _same(got, ref), ([g.tokens for g in got], [r["tokens"] for r in ref])
        outs.append([g.tokens for g in got])
    fp.set_no_context(True); op.no_context = 1
    st.full(fp, b)
    clean = [g.tokens for g in st.batch_segments(0)]
    assert _same(st.batch_segments(0), ost.full(op, b))
    assert clean != outs[1] or clean != outs[2], "the carried text context never changed a token"
    # the flag clears the context at the START of a call only: a call with no_context = false after that one is conditioned on
    # its text, and the call after it on both
    fp.set_no_context(False); op.no_context = 0
    for pcm in (a, b):
        st.full(fp, pcm)
        ref = ost.full(op, pcm)
        assert _same(st.batch_segments(0), ref), ([g.tokens for g in st.batch_segments(0)], [r["tokens"] for r in ref])
    st.free(); ctx.free(); ost.close(); om.close()


def test_one_bad_utterance_does_not_fail_the_batch(wr, model_path):
    """Language detection on a clip too short to hold a frame fails that utterance (code -3) -- and only that one: the others
    in the batch return what they return alone. Out-of-vocabulary prompt tokens are rejected before anything runs."""
    path = model_path("micro", decisive=4.0)
    ctx = wr.WhisperContext(path)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("auto")
    pcms = [synth_pcm(700, 10 * 16000), synth_pcm(701, 30), synth_pcm(702, 5 * 16000)]
    st = ctx.create_batch_state(3, 1, 2)
    assert st.full_batch(fp, pcms, allow_failed=True) == [0, -3, 0]
    assert [st.batch_rc(u) for u in range(3)] == [0, -3, 0]
    with pytest.raises(wr.WhisperError):          # without the opt-in a failed utterance is an error, not an empty transcript
        st.full_batch(fp, pcms)
    assert st.batch_tokens(1) == [] and len(st.batch_tokens(0)) > 0 and len(st.batch_tokens(2)) > 0
    one = ctx.create_state()
    for u in (0, 2):
        one.full(fp, pcms[u])
        assert one.batch_tokens(0) == st.batch_tokens(u)
    with pytest.raises(wr.WhisperError):
        one.full(fp, pcms[1])                       # the one-utterance call reports its utterance's failure
    # the manager front hands the failure to the caller of that clip only
    from handy_amd.manager import TranscriptionManager
    import threading
    tm = TranscriptionManager(max_batch=4, collect_ms=100.0, selected_language="auto")
    tm.load_model(path)
    res = [None] * 3
    def call(i):
        try: res[i] = tm.transcribe(pcms[i])
        except RuntimeError as e: res[i] = e
    th = [threading.Thread(target=call, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    tm.unload_model()
    assert isinstance(res[1], RuntimeError) and isinstance(res[0], str) and isinstance(res[2], str)
    fp2 = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp2.set_language("en"); fp2.set_tokens([5, 51865, 7])
    with pytest.raises(wr.WhisperError):
        st.full_batch(fp2, pcms)
    fp2.set_tokens([5, -1])
    with pytest.raises(wr.WhisperError):
        st.full_batch(fp2, pcms)
    st.free(); one.free(); ctx.free()


@pytest.mark.parametrize("beam", [1, 3])
def test_pass_graphs_do_not_change_results(wr, model_path, beam):
    """Decoder passes of a repeated shape are replayed as HIP graphs (captured the second time a shape comes by). Tokens,
    probabilities and segment times must be those of plain launches, bit for bit: a ragged greedy batch (shapes change as
    utterances finish), beam search (page copies and row groups vary from pass to pass) and one clip step by step."""
    path = model_path("micro", decisive=4.0)
    ctx = wr.WhisperContext(path)
    fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(beam, -1.0) if beam > 1 else wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0); fp.set_language("en")
    pcms = [synth_pcm(900 + i, int(16000 * s)) for i, s in enumerate((30.0, 6.0, 14.5, 30.0, 2.5, 21.0))]
    outs = {}
    for graphs in (False, True):
        st = ctx.create_batch_state(len(pcms), beam, 4)
        st.set_pass_graphs(graphs)
        st.set_beam_candidates_sampled(False)
        runs = []
        for _ in range(2):                               # the second call replays what the first one captured
            st.full_batch(fp, pcms)
            runs.append([[(g.t0, g.t1, g.tokens, g.plog) for g in st.batch_segments(u)] for u in range(len(pcms))])
        one = []
        for pcm in pcms[:2]:
            st.full_batch(fp, [pcm])
            one.append([(g.t0, g.t1, g.tokens) for g in st.batch_segments(0)])
        assert runs[0] == runs[1]
        outs[graphs] = (runs[0], one)
        st.free()
    ctx.free()
    assert outs[True] == outs[False]
    assert sum(len(g[2]) for u in outs[True][0] for g in u) > 50


def test_node_driver_one_gpu(wr, model_path, monkeypatch):
    """wmi_node_run on the one GPU of the test box: one engine; then two engines (two worker threads) on that same GPU, which
    share the batch queue; then one engine with the records forced through a one-rank RCCL communicator (the library loads
    librccl at that moment). Records must say what wmi_full_batch says for every clip, in utterance order, a failing clip
    carrying its own code. (More than one distinct GPU -- the send / receive of the gather -- has not run anywhere.)"""
    from handy_amd.node import NativeNode
    path = model_path("micro", decisive=4.0)
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("auto")
    secs = (30.0, 6.0, 14.5, 2.0, 21.0, 9.0, 30.0, 3.5, 12.0, 5.0, 17.0)
    pcms = [synth_pcm(950 + i, int(16000 * s)) for i, s in enumerate(secs)]
    pcms[4] = synth_pcm(999, 30)                     # too short for language detection: fails alone (-3)
    ctx = wr.WhisperContext(path)
    st = ctx.create_batch_state(len(pcms), 1, 4)
    codes = st.full_batch(fp, pcms, allow_failed=True)
    want = [(codes[u], st.batch_tokens(u), st.batch_lang_id(u) if codes[u] == 0 else None) for u in range(len(pcms))]
    st.free(); ctx.free()
    assert codes[4] == -3 and sum(len(t) for _, t, _ in want) > 50
    for devices, gather in (([0], "host"), ([0, 0], "host"), ([0], "rccl")):
        monkeypatch.setenv("WMI_NODE_GATHER", gather)
        node = NativeNode(path, devices, batch=3, max_decoders=1, enc_chunk=2)
        recs = node.run(fp, pcms)
        assert node.last_gather() == gather, (devices, gather, node.last_gather())
        assert sum(node.batches_per_engine()) == 4
        for u, r in enumerate(recs):
            assert r.utt == u and r.rc == want[u][0] and 0 <= r.device < len(devices)
            assert list(r.tokens[: r.n_tokens]) == want[u][1], (devices, u)
            if r.rc == 0:
                assert r.lang_id == want[u][2] and r.n_segments >= 1 and r.t1 > r.t0 >= 0
        node.free()
