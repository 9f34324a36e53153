"""Host-side sanitizer runs (CPU only; the GPU build is never instrumented).

* the ggml file parser of the engine (handy_amd/csrc/model_parse.cpp) built with gcc under AddressSanitizer +
  UndefinedBehaviorSanitizer and driven over truncated, bit-flipped and oversized-header images of a valid model file:
  Handy's downloader resumes partial files and only compares sizes before the loader gets them (reference:
  src-tauri/src/managers/model.rs:361-405, 496-513), so truncated and corrupt files do reach `whisper_init_*`;
* the CPU checker itself (oracle/whisper_oracle.c, `make -C oracle SAN=1`) through log-mel, encoder, decoder and whisper_full;
* the batching manager front under Python's own thread checks is covered by tests/test_cpu_host.py.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "handy_amd", "csrc")
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98",
               OMP_NUM_THREADS="4")
SAN_ENV.pop("LD_PRELOAD", None)          # the sanitizer runtime must come first in the library list


@pytest.fixture(scope="module")
def micro_model(tmp_path_factory):
    from handy_amd import ggml_file
    p = str(tmp_path_factory.mktemp("san") / "micro.bin")
    ggml_file.write_synthetic_model(p, "micro", seed=3, decisive=4.0)
    return p


def _run_parser(model, mutations):
    subprocess.check_call(["make", "-C", CSRC, "parse_san"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(CSRC, "build_san", "parse_san"), model], input="\n".join(mutations) + "\n", text=True,
                       capture_output=True, env=SAN_ENV, timeout=600)
    assert r.returncode == 0, f"sanitizer or crash (exit {r.returncode}):\n{r.stderr[-3000:]}"
    out = {}
    for line in r.stdout.splitlines():
        k, v = line.split(" -> ")
        out[k] = v
    assert len(out) == len(set(mutations)), (len(out), len(mutations))
    return out


def test_parser_under_sanitizers_on_truncated_and_corrupt_files(micro_model):
    size = os.path.getsize(micro_model)
    img = open(micro_model, "rb").read()
    n_mel = struct.unpack_from("<i", img, 4 + 9 * 4)[0]
    off_mel = 4 + 11 * 4                       # int32 n_mel, int32 n_fft, then the filter table
    off_vocab = off_mel + 8 + n_mel * 201 * 4   # int32 token count, then {u32 len, bytes} per token
    n_tok = struct.unpack_from("<i", img, off_vocab)[0]
    p = off_vocab + 4
    for _ in range(n_tok):
        p += 4 + struct.unpack_from("<I", img, p)[0]
    off_t0 = p                                 # first tensor header: n_dims, name_len, type, ne[n_dims], name, data
    n_dims = struct.unpack_from("<i", img, off_t0)[0]
    assert 1 <= n_dims <= 4 and off_t0 < size

    rng = np.random.default_rng(0)
    muts = ["valid"]
    # truncations: inside the header, the filter table, the vocabulary, the first tensor header, tensor data, the last byte
    cuts = [0, 1, 3, 4, 20, 47, 48, off_mel + 3, off_mel + 8, off_mel + 8 + 1000, off_vocab, off_vocab + 3, off_vocab + 4, off_vocab + 7,
            off_t0 - 1, off_t0, off_t0 + 4, off_t0 + 11, off_t0 + 12 + 4 * n_dims, off_t0 + 12 + 4 * n_dims + 5, size - 1, size // 2, size // 3]
    cuts += [int(v) for v in rng.integers(0, size, 40)]
    muts += [f"trunc {c}" for c in sorted(set(cuts))]
    # every hyper-parameter with hostile values (0, -1, huge, just off)
    for i in range(11):
        for v in (0, -1, 0x7fffffff, 0x80000000, 1, 65, 51863, 100000):
            muts.append(f"set32 {4 + 4 * i} {v}")
    # the filter table's dimensions, the token count, a token length, the first tensor's header fields
    for off in (off_mel, off_mel + 4, off_vocab, off_vocab + 4, off_t0, off_t0 + 4, off_t0 + 8, off_t0 + 12, off_t0 + 12 + 4 * (n_dims - 1)):
        for v in (0, -1, 0x7fffffff, 0x40000000, 5, 129, 4097, 0xffffffff):
            muts.append(f"set32 {off} {v}")
    # random bit flips in the structural part of the file and anywhere
    muts += [f"flip {int(o)} {int(b)}" for o, b in zip(rng.integers(0, off_vocab + 64, 60), rng.integers(0, 8, 60))]
    muts += [f"flip {int(o)} {int(b)}" for o, b in zip(rng.integers(off_t0, min(size, off_t0 + 4096), 60), rng.integers(0, 8, 60))]
    muts = list(dict.fromkeys(muts))
    out = _run_parser(micro_model, muts)
    assert out["valid"].startswith("ok")
    for m in muts:
        if m.startswith("trunc") and int(m.split()[1]) < size:
            assert out[m] == "rejected", (m, out[m])           # a file cut anywhere is refused, not read past its end
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):                       # a zero, negative or huge hyper-parameter is refused
        for v in (0, -1, 0x7fffffff, 0x80000000):
            assert out[f"set32 {4 + 4 * i} {v}"] == "rejected", (i, v)
    for v in (0, -1, 0x7fffffff, 0x40000000, 0xffffffff):          # so is a hostile tensor rank or name length
        assert out[f"set32 {off_t0} {v}"] == "rejected" and out[f"set32 {off_t0 + 4} {v}"] == "rejected"
    assert sum(v == "rejected" for v in out.values()) > 150


def test_oracle_under_sanitizers(micro_model):
    """The CPU checker is C with hand-written indexing: run its whole path once under ASan + UBSan."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "SAN=1"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_san"), micro_model, "2"], text=True, capture_output=True, env=SAN_ENV, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok "), f"exit {r.returncode}\n{r.stdout}\n{r.stderr[-3000:]}"
    assert int(r.stdout.split()[1]) > 0
    # a truncated file is refused by the checker's loader as well
    cut = micro_model + ".cut"
    with open(micro_model, "rb") as f, open(cut, "wb") as g:
        g.write(f.read(os.path.getsize(micro_model) // 2))
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_san"), cut], text=True, capture_output=True, env=SAN_ENV, timeout=300)
    assert r.returncode == 3, f"exit {r.returncode}\n{r.stdout}\n{r.stderr[-3000:]}"


def test_tokenizer_follows_the_word_pattern(micro_model):
    """`whisper_tokenize` / `initial_prompt`: whisper.cpp cuts the text into words with the GPT-2 pre-tokenisation pattern and
    covers each word with the longest vocabulary entries (no merge ranks). The engine scans that pattern by hand
    (model_parse.cpp); here the same pattern goes through Python's `re` on bytes and the two must agree token for token, on text
    with contractions, digits, punctuation, white-space runs, tabs, UTF-8 and words that only exist with a leading space.
    Parity unpinned: whisper.cpp's source is not in the reference tree, the pattern and the loop are as recalled."""
    import re
    from handy_amd import ggml_file
    m = ggml_file.read_model(micro_model) if hasattr(ggml_file, "read_model") else None
    vocab = m.vocab if m is not None else ggml_file.synthetic_vocab(ggml_file.n_file_tokens(ggml_file.hparams_for("micro")))
    ids = {}
    for i, t in enumerate(vocab):
        ids[t] = i
    pat = re.compile(rb"'s|'t|'re|'ve|'m|'ll|'d| ?[A-Za-z]+| ?[0-9]+| ?[^\sA-Za-z0-9]+|\s+(?!\S)|\s+")

    def expected(text):
        out = []
        for w in pat.findall(text):
            i = 0
            while i < len(w):
                for j in range(len(w), i, -1):
                    if w[i:j] in ids:
                        out.append(ids[w[i:j]]); i = j
                        break
                else:
                    i += 1
        return out

    words = [t for t in vocab[256:400]]
    rng = np.random.default_rng(5)
    texts = [b"", b" ", b"  ", b"a", b"it's the kato's", b"we'll they've I'm you'd can't 're", b"12 345  6789", b"  lead and trail  ", b"tab\tsep\n\nnew  \t line",
             b"semi;colon -- dash...", "café naïve 日本".encode(), b"'", b"''s", b"x'sy", b" 'll", b"a  b   c    d", b"end with space \t"]
    for _ in range(40):
        parts = []
        for _ in range(int(rng.integers(1, 9))):
            k = int(rng.integers(0, 6))
            parts.append([words[int(rng.integers(len(words)))], words[int(rng.integers(len(words)))].strip() + b"'s", str(int(rng.integers(0, 99999))).encode(),
                          bytes(rng.choice(list(b" \t\n.,;!?-'\""), int(rng.integers(1, 4))).tolist()), b" ", bytes([int(rng.integers(128, 256))])][k])
        texts.append(b"".join(parts))
    muts = list(dict.fromkeys(("tok " + t.hex()).strip() for t in texts))
    out = _run_parser(micro_model, muts)
    some = 0
    for t in texts:
        got = [int(x) for x in out[("tok " + t.hex()).strip()].split()]
        assert got[0] == len(got) - 1
        got = got[1:]
        assert got == expected(t), (t, got, expected(t))
        some += len(got) > 3
    assert some > 30
