"""Legacy ggml Whisper ``.bin`` container: writer, reader and a seeded synthetic-model generator.

The Handy model catalog ships Whisper weights in this container
(reference: src-tauri/src/managers/model.rs:71-146 -- ggml-small.bin f16,
whisper-medium-q4_1.bin Q4_1, ggml-large-v3-turbo.bin f16, ggml-large-v3-q5_0.bin Q5_0),
so the engine's loader and every test model use the same byte layout:

    u32 magic 0x67676d6c
    11 x i32  n_vocab n_audio_ctx n_audio_state n_audio_head n_audio_layer
              n_text_ctx n_text_state n_text_head n_text_layer n_mels ftype
    i32 n_mel, i32 n_fft, f32[n_mel*n_fft] mel filters
    i32 n_tokens, then per token {u32 len, bytes}
    tensors until EOF: {i32 n_dims, i32 name_len, i32 ttype, i32 ne[n_dims] (fastest first),
                        name bytes, raw data}

No weights ship with the reference and none can be fetched, so tests and the
benchmark run on seeded synthetic files of the exact architecture written here.
This module is host-side tooling (numpy only); it does no inference.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Tuple

import numpy as np

GGML_MAGIC = 0x67676D6C

# ggml tensor type ids used by whisper files
T_F32, T_F16, T_Q4_0, T_Q4_1, T_Q5_0, T_Q5_1, T_Q8_0 = 0, 1, 2, 3, 6, 7, 8
# hparams.ftype -> tensor type of the 2-D weight matrices
FTYPE_TO_TTYPE = {0: T_F32, 1: T_F16, 2: T_Q4_0, 3: T_Q4_1, 7: T_Q8_0, 8: T_Q5_0, 9: T_Q5_1}
BLOCK = 32
TTYPE_BLOCK_BYTES = {T_Q4_0: 18, T_Q4_1: 20, T_Q5_0: 22, T_Q5_1: 24, T_Q8_0: 34}

ARCHS = {
    # name: (n_vocab, n_state, n_head, n_enc_layer, n_dec_layer, n_mels)
    "tiny.en": (51864, 384, 6, 4, 4, 80),
    "tiny": (51865, 384, 6, 4, 4, 80),
    "base": (51865, 512, 8, 6, 6, 80),
    "small": (51865, 768, 12, 12, 12, 80),
    "medium": (51865, 1024, 16, 24, 24, 80),
    "large-v3": (51866, 1280, 20, 32, 32, 128),
    "large-v3-turbo": (51866, 1280, 20, 32, 4, 128),
    # test-sized architectures (real vocabulary size so the special-token ids stay real)
    # (3 decoder layers: upstream treats 2-layer decoders with these vocabularies as distilled
    # models and forces no_timestamps, which would hide the timestamp rules from the tests)
    "micro.en": (51864, 128, 2, 2, 3, 80),
    "micro": (51865, 128, 2, 2, 3, 80),
    "micro-v3": (51866, 128, 2, 2, 3, 128),
    # tiny's width with two layers each: wide enough for the decoder's split-K GEMM path (K = 384 = 6 K-steps)
    "mini": (51865, 384, 6, 2, 2, 80),
    # base's and small's widths with few layers: the widths at which the decoder's QKV (d < 853) and fc1 (d < 640) projections
    # take the split-K path as well
    "mini512": (51865, 512, 8, 2, 3, 80),
    "mini768": (51865, 768, 12, 2, 3, 80),
}


@dataclass
class HParams:
    n_vocab: int
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4
    n_mels: int = 80
    ftype: int = 1

    def as_list(self) -> List[int]:
        return [self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head,
                self.n_audio_layer, self.n_text_ctx, self.n_text_state, self.n_text_head,
                self.n_text_layer, self.n_mels, self.ftype]


def hparams_for(arch: str, ftype: int = 1) -> HParams:
    v, d, h, le, ld, mel = ARCHS[arch]
    return HParams(n_vocab=v, n_audio_state=d, n_audio_head=h, n_audio_layer=le,
                   n_text_state=d, n_text_head=h, n_text_layer=ld, n_mels=mel, ftype=ftype)


# --------------------------------------------------------------------------------------
# mel filterbank (librosa "slaney" scale + slaney area normalisation, fmin 0, fmax 8000)
# --------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_mels: int, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[n_mels, n_fft/2+1] float32 table, the same construction as OpenAI's mel_filters.npz."""
    n_bins = n_fft // 2 + 1
    fftfreqs = np.linspace(0, sr / 2, n_bins)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


# --------------------------------------------------------------------------------------
# vocabulary
# --------------------------------------------------------------------------------------
def _gpt2_byte_order() -> List[int]:
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in bs]
    return bs + rest


def synthetic_vocab(n_file_tokens: int) -> List[bytes]:
    """First 256 tokens are the single bytes in GPT-2 byte-level order (so token 220 is b" ",
    as in the real Whisper vocabularies); the remaining ones are unique ASCII pseudo-words."""
    toks: List[bytes] = [bytes([b]) for b in _gpt2_byte_order()]
    syll = [b"ka", b"to", b"mi", b"re", b"su", b"no", b"la", b"de", b"vi", b"po", b"en", b"ar", b"ul", b"is", b"om", b"ti"]
    i = 256
    while len(toks) < n_file_tokens:
        n = i
        w = b""
        while True:
            w += syll[n & 15]
            n >>= 4
            if n == 0:
                break
        toks.append((b" " if (i & 1) else b"") + w)
        i += 1
    return toks[:n_file_tokens]


# --------------------------------------------------------------------------------------
# block quantisers (ggml reference semantics) -- used only to WRITE test files
# --------------------------------------------------------------------------------------
def _f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def quantize(x: np.ndarray, ttype: int) -> bytes:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, BLOCK)
    nb = x.shape[0]
    if ttype == T_Q8_0:
        amax = np.abs(x).max(axis=1)
        d = amax / 127.0
        idv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0.0).astype(np.float32)
        q = np.round(x * idv[:, None]).astype(np.int8)
        out = np.zeros((nb, 34), dtype=np.uint8)
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 2:] = q.view(np.uint8)
        return out.tobytes()
    if ttype in (T_Q4_0, T_Q5_0):
        bits = 4 if ttype == T_Q4_0 else 5
        idx = np.abs(x).argmax(axis=1)
        mx = x[np.arange(nb), idx]
        d = mx / -(1 << (bits - 1))
        idv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0.0).astype(np.float32)
        off = float(1 << (bits - 1)) + 0.5
        q = np.minimum((1 << bits) - 1, (x * idv[:, None] + off).astype(np.int32).clip(0)).astype(np.uint8)
    elif ttype in (T_Q4_1, T_Q5_1):
        bits = 4 if ttype == T_Q4_1 else 5
        mn = x.min(axis=1)
        mxv = x.max(axis=1)
        d = (mxv - mn) / ((1 << bits) - 1)
        idv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0.0).astype(np.float32)
        q = np.minimum((1 << bits) - 1, ((x - mn[:, None]) * idv[:, None] + 0.5).astype(np.int32).clip(0)).astype(np.uint8)
    else:
        raise ValueError(f"unsupported ttype {ttype}")
    lo, hi = q[:, :16], q[:, 16:]
    qs = ((lo & 0x0F) | ((hi & 0x0F) << 4)).astype(np.uint8)
    parts = [_f16(d).view(np.uint8).reshape(nb, 2)]
    if ttype in (T_Q4_1, T_Q5_1):
        parts.append(_f16(mn).view(np.uint8).reshape(nb, 2))
    if ttype in (T_Q5_0, T_Q5_1):
        qh = np.zeros(nb, dtype=np.uint32)
        for j in range(16):
            qh |= ((lo[:, j].astype(np.uint32) >> 4) & 1) << j
            qh |= ((hi[:, j].astype(np.uint32) >> 4) & 1) << (j + 16)
        parts.append(qh.view(np.uint8).reshape(nb, 4))
    parts.append(qs)
    return np.concatenate(parts, axis=1).tobytes()


def dequantize(raw: bytes, ttype: int, n: int) -> np.ndarray:
    bb = TTYPE_BLOCK_BYTES[ttype]
    nb = n // BLOCK
    a = np.frombuffer(raw, dtype=np.uint8, count=nb * bb).reshape(nb, bb)
    d = a[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb)
    p = 2
    m = None
    if ttype in (T_Q4_1, T_Q5_1):
        m = a[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb)
        p = 4
    if ttype == T_Q8_0:
        q = a[:, 2:].copy().view(np.int8).astype(np.float32)
        return (q * d[:, None]).reshape(-1)
    qh = None
    if ttype in (T_Q5_0, T_Q5_1):
        qh = a[:, p:p + 4].copy().view(np.uint32).reshape(nb)
        p += 4
    qs = a[:, p:p + 16]
    lo = (qs & 0x0F).astype(np.int32)
    hi = (qs >> 4).astype(np.int32)
    if qh is not None:
        for j in range(16):
            lo[:, j] |= ((qh >> j) & 1).astype(np.int32) << 4
            hi[:, j] |= ((qh >> (j + 16)) & 1).astype(np.int32) << 4
    q = np.concatenate([lo, hi], axis=1).astype(np.float32)
    if ttype == T_Q4_0:
        return ((q - 8.0) * d[:, None]).reshape(-1)
    if ttype == T_Q5_0:
        return ((q - 16.0) * d[:, None]).reshape(-1)
    return (q * d[:, None] + m[:, None]).reshape(-1)


# --------------------------------------------------------------------------------------
# tensor naming
# --------------------------------------------------------------------------------------
def tensor_specs(hp: HParams) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, numpy shape) in file order. numpy shape is slowest-first, i.e. reversed ggml ne."""
    d, dt = hp.n_audio_state, hp.n_text_state
    out: List[Tuple[str, Tuple[int, ...]]] = []
    out += [("encoder.positional_embedding", (hp.n_audio_ctx, d)),
            ("encoder.conv1.weight", (d, hp.n_mels, 3)), ("encoder.conv1.bias", (d, 1)),
            ("encoder.conv2.weight", (d, d, 3)), ("encoder.conv2.bias", (d, 1))]
    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}."
        out += [(p + "attn_ln.weight", (d,)), (p + "attn_ln.bias", (d,)),
                (p + "attn.query.weight", (d, d)), (p + "attn.query.bias", (d,)),
                (p + "attn.key.weight", (d, d)),
                (p + "attn.value.weight", (d, d)), (p + "attn.value.bias", (d,)),
                (p + "attn.out.weight", (d, d)), (p + "attn.out.bias", (d,)),
                (p + "mlp_ln.weight", (d,)), (p + "mlp_ln.bias", (d,)),
                (p + "mlp.0.weight", (4 * d, d)), (p + "mlp.0.bias", (4 * d,)),
                (p + "mlp.2.weight", (d, 4 * d)), (p + "mlp.2.bias", (d,))]
    out += [("encoder.ln_post.weight", (d,)), ("encoder.ln_post.bias", (d,))]
    out += [("decoder.positional_embedding", (hp.n_text_ctx, dt)),
            ("decoder.token_embedding.weight", (hp.n_vocab, dt))]
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}."
        out += [(p + "attn_ln.weight", (dt,)), (p + "attn_ln.bias", (dt,)),
                (p + "attn.query.weight", (dt, dt)), (p + "attn.query.bias", (dt,)),
                (p + "attn.key.weight", (dt, dt)),
                (p + "attn.value.weight", (dt, dt)), (p + "attn.value.bias", (dt,)),
                (p + "attn.out.weight", (dt, dt)), (p + "attn.out.bias", (dt,)),
                (p + "cross_attn_ln.weight", (dt,)), (p + "cross_attn_ln.bias", (dt,)),
                (p + "cross_attn.query.weight", (dt, dt)), (p + "cross_attn.query.bias", (dt,)),
                (p + "cross_attn.key.weight", (dt, dt)),
                (p + "cross_attn.value.weight", (dt, dt)), (p + "cross_attn.value.bias", (dt,)),
                (p + "cross_attn.out.weight", (dt, dt)), (p + "cross_attn.out.bias", (dt,)),
                (p + "mlp_ln.weight", (dt,)), (p + "mlp_ln.bias", (dt,)),
                (p + "mlp.0.weight", (4 * dt, dt)), (p + "mlp.0.bias", (4 * dt,)),
                (p + "mlp.2.weight", (dt, 4 * dt)), (p + "mlp.2.bias", (dt,))]
    out += [("decoder.ln.weight", (dt,)), ("decoder.ln.bias", (dt,))]
    return out


def _stays_f32(name: str, shape: Tuple[int, ...]) -> bool:
    """With a 16-bit or quantised ftype, 1-D tensors, the [d,1] conv biases and both
    positional embeddings stay f32 (what the upstream converter writes)."""
    if len(shape) < 2 or shape[-1] == 1:
        return True
    return name in ("encoder.positional_embedding", "decoder.positional_embedding")


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2))
    t = np.arange(length)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def special_token_ids(n_vocab: int) -> Dict[str, int]:
    """eot / sot / beg ... of a Whisper vocabulary of this size (English-only 51864, multilingual 51865, 51866)."""
    if n_vocab < 51865:
        return dict(eot=50256, sot=50257, translate=50357, transcribe=50358, solm=50359, prev=50360, nosp=50361, tok_not=50362, beg=50363)
    dt = n_vocab - 51765 - 1 - 98
    return dict(eot=50257, sot=50258, translate=50357 + dt, transcribe=50358 + dt, solm=50359 + dt, prev=50360 + dt, nosp=50361 + dt,
                tok_not=50362 + dt, beg=50363 + dt)


def synthetic_tensors(hp: HParams, seed: int = 0, decisive: float = 1.0, style: str = "plain") -> Iterable[Tuple[str, np.ndarray]]:
    """Seeded random weights sized so activations stay O(1) through the stack.
    `decisive` scales the token embedding (hence the logits' spread).
    style "plain": every tensor an independent Gaussian. With the output projection tied to the token embedding, a decoded
    token's own embedding then dominates the next step's logits and greedy decoding settles on one repeated token.
    style "varied": the decoder's positional embedding and its attention outputs outweigh the token embedding, the cross-
    attention is sharp, decoder biases are zero and the timestamp / end-of-text rows of the embedding are a little longer,
    so that decoding walks through many different text tokens, timestamps and segment ends, and depends on the audio --
    the control flow the whisper_full rules exist for. Test weights either way; nothing here resembles a trained model."""
    rng = np.random.default_rng(seed)
    ids = special_token_ids(hp.n_vocab)
    for name, w in _plain_tensors(hp, rng, decisive):
        if style == "varied":
            if name == "decoder.token_embedding.weight":
                w = w.copy(); w[ids["beg"]:] *= 1.1; w[ids["eot"]] *= 2.0
            elif name == "decoder.positional_embedding":
                w = w * 30.0
            elif name == "decoder.ln.bias":
                w = w * 0.0
            elif name.startswith("decoder.blocks."):
                if name.endswith(".bias") and not name.endswith("_ln.bias"):
                    w = w * 0.0
                elif name.endswith("cross_attn.out.weight"):
                    w = w * 6.0
                elif name.endswith(".attn.out.weight"):
                    w = w * 2.0
                elif name.endswith("cross_attn.query.weight") or name.endswith("cross_attn.key.weight"):
                    w = w * 4.0
        elif style != "plain":
            raise ValueError(f"unknown synthetic weight style {style!r}")
        yield name, w.astype(np.float32)


def _plain_tensors(hp: HParams, rng, decisive: float) -> Iterable[Tuple[str, np.ndarray]]:
    for name, shape in tensor_specs(hp):
        if name == "encoder.positional_embedding":
            w = sinusoids(shape[0], shape[1])
        elif name == "decoder.positional_embedding":
            w = rng.standard_normal(shape, dtype=np.float32) * 0.05
        elif name == "decoder.token_embedding.weight":
            w = rng.standard_normal(shape, dtype=np.float32) * (0.06 * decisive)
        elif name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
            w = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith(".bias"):
            w = 0.05 * rng.standard_normal(shape, dtype=np.float32)
        elif "conv" in name:
            fan_in = shape[1] * shape[2]
            w = rng.standard_normal(shape, dtype=np.float32) * (1.2 / np.sqrt(fan_in))
        else:
            fan_in = shape[-1]
            g = 1.4 if (".attn.query" in name or ".attn.key" in name or "cross_attn.query" in name or "cross_attn.key" in name) else 0.8
            w = rng.standard_normal(shape, dtype=np.float32) * (g / np.sqrt(fan_in))
        yield name, w.astype(np.float32)


# --------------------------------------------------------------------------------------
# writer / reader
# --------------------------------------------------------------------------------------
def write_header(f, hp: HParams, filters: np.ndarray, vocab: List[bytes]) -> None:
    f.write(struct.pack("<I", GGML_MAGIC))
    f.write(struct.pack("<11i", *hp.as_list()))
    f.write(struct.pack("<ii", filters.shape[0], filters.shape[1]))
    f.write(np.ascontiguousarray(filters, dtype="<f4").tobytes())
    f.write(struct.pack("<i", len(vocab)))
    for t in vocab:
        f.write(struct.pack("<I", len(t)))
        f.write(t)


def write_tensor(f, name: str, arr: np.ndarray, ttype: int) -> None:
    nb = name.encode()
    ne = list(reversed(arr.shape))
    f.write(struct.pack("<iii", len(ne), len(nb), ttype))
    f.write(struct.pack(f"<{len(ne)}i", *ne))
    f.write(nb)
    if ttype == T_F32:
        f.write(np.ascontiguousarray(arr, dtype="<f4").tobytes())
    elif ttype == T_F16:
        f.write(np.ascontiguousarray(arr, dtype=np.float32).astype("<f2").tobytes())
    else:
        f.write(quantize(arr, ttype))


def n_file_tokens(hp: HParams) -> int:
    return 50257 if hp.n_vocab >= 51865 else 50256


def write_synthetic_model(path: str, arch: str = "micro.en", seed: int = 0, ftype: int = 1,
                          decisive: float = 1.0, hp: HParams | None = None, filters: np.ndarray | None = None, style: str = "plain") -> HParams:
    """`filters` replaces the standard [n_mels, 201] mel table (tests of unusual tables); `style`: see synthetic_tensors."""
    hp = hp or hparams_for(arch, ftype)
    hp.ftype = ftype
    wtype = FTYPE_TO_TTYPE[ftype % 1000]
    with open(path, "wb") as f:
        write_header(f, hp, mel_filterbank(hp.n_mels) if filters is None else filters, synthetic_vocab(n_file_tokens(hp)))
        for name, w in synthetic_tensors(hp, seed, decisive, style):
            tt = T_F32 if _stays_f32(name, w.shape) else wtype
            if tt not in (T_F32, T_F16) and w.shape[-1] % BLOCK != 0:
                tt = T_F16
            write_tensor(f, name, w, tt)
    return hp


@dataclass
class GgmlModel:
    hp: HParams
    filters: np.ndarray
    vocab: List[bytes]
    tensors: Dict[str, np.ndarray] = field(default_factory=dict)   # dequantised to f32
    ttypes: Dict[str, int] = field(default_factory=dict)


def read_model(path: str) -> GgmlModel:
    with open(path, "rb") as f:
        data = f.read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, data, off)
        off += struct.calcsize(fmt)
        return v

    (magic,) = take("<I")
    if magic != GGML_MAGIC:
        raise ValueError("bad magic")
    hp = HParams(*take("<11i"))
    n_mel, n_fft = take("<ii")
    filters = np.frombuffer(data, dtype="<f4", count=n_mel * n_fft, offset=off).reshape(n_mel, n_fft).copy()
    off += 4 * n_mel * n_fft
    (nv,) = take("<i")
    vocab = []
    for _ in range(nv):
        (ln,) = take("<I")
        vocab.append(data[off:off + ln])
        off += ln
    m = GgmlModel(hp, filters, vocab)
    while off < len(data):
        n_dims, name_len, ttype = take("<iii")
        ne = take(f"<{n_dims}i")
        name = data[off:off + name_len].decode()
        off += name_len
        n = int(np.prod(ne))
        shape = tuple(reversed(ne))
        if ttype == T_F32:
            arr = np.frombuffer(data, dtype="<f4", count=n, offset=off).astype(np.float32)
            off += 4 * n
        elif ttype == T_F16:
            arr = np.frombuffer(data, dtype="<f2", count=n, offset=off).astype(np.float32)
            off += 2 * n
        else:
            nbytes = n // BLOCK * TTYPE_BLOCK_BYTES[ttype]
            arr = dequantize(data[off:off + nbytes], ttype, n)
            off += nbytes
        m.tensors[name] = arr.reshape(shape)
        m.ttypes[name] = ttype
    return m
