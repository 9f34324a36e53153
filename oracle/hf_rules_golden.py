"""Independent pins for the whisper_full decoding rules, from Hugging Face transformers (runs only in the build
container; writes tests/golden/hf_rules.npz and tests/golden/hf_sequences.npz).

Part A -- logits rules: hand-built logits and token contexts through transformers' own processors
  SuppressTokensLogitsProcessor, SuppressTokensAtBeginLogitsProcessor, WhisperTimeStampLogitsProcessor
  (transformers/generation/logits_process.py:1816-2047); stored per case: which entries stay finite, the argmax and its
  log-probability. The logits themselves are regenerated from the recipe in rules_case_logits().
Part B -- greedy decoding with timestamps: WhisperForConditionalGeneration.generate() on seeded synthetic models
  ("varied" weights) and clips shorter than 30 s; stored: the generated token ids.

Where whisper.cpp's rules differ from transformers' (recalled from whisper.cpp's whisper_process_logits; not in the reference
tree), the difference is named here and the tests apply it explicitly through logits_filter_callback (hf_delta_filter):
  D1  first sampled position: transformers (like OpenAI's decoding.py) masks every text token, so the first token is a
      timestamp; whisper.cpp does not (it only bounds the first timestamp by max_initial_ts and relies on the
      timestamp-mass rule).
  D2  increasing timestamps: transformers forbids timestamps below the last one, and the last one itself unless it closes
      a segment (last token a timestamp, the one before it text); whisper.cpp forbids timestamps below the last one that is
      greater than <|0.00|> (has_ts / seek_delta), never the last one itself.
  D3  (numbers only, never the choice) after the timestamp-mass rule has masked the text tokens whisper.cpp keeps the
      log-probabilities of the softmax taken before it; transformers' sampler renormalises over what is left.
Usage: HF_HUB_OFFLINE=1 python oracle/hf_rules_golden.py [--write]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from handy_amd import ggml_file as gf          # noqa: E402
from handy_amd.synth import synth_pcm          # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
VOCABS = {"micro.en": 51864, "micro": 51865, "micro-v3": 51866}
SEQ_CLIPS = ((21, 29.0), (22, 12.5), (23, 20.0), (24, 6.0), (25, 25.5), (26, 16.0))
N_MAX = 448 // 2 - 4          # whisper.cpp's cap on sampled tokens per window


def suppress_list(V: int):
    """The ids whisper.cpp always masks: <|notimestamps|>, <|startoftranscript|>, <|nospeech|>, <|startoflm|>, the task and
    <|startofprev|> tokens, every language token."""
    ids = gf.special_token_ids(V)
    out = [ids["tok_not"], ids["sot"], ids["nosp"], ids["solm"], ids["translate"], ids["transcribe"], ids["prev"]]
    out += [ids["sot"] + 1 + i for i in range(100)]        # English-only vocabularies hold the language tokens too
    return sorted(set(out))


def n_prompt(V: int) -> int:
    return 3 if V >= 51865 else 1           # [sot, language, task] or [sot]


def rules_cases(V: int):
    """(name, context tokens, recipe, applies) -- `applies`: which documented difference, if any, shows in this case."""
    ids = gf.special_token_ids(V)
    beg, eot = ids["beg"], ids["eot"]
    T = lambda k: beg + k
    return [
        ("initial_text_dominant", [], dict(seed=1, bumps={1000: 14.0, T(20): 6.0, 220: 13.0, eot: 13.5}), "D1"),
        ("initial_timestamp_mass", [], dict(seed=2, scale=0.5, ts_shift=6.0, bumps={1000: 9.0, T(70): 12.0, T(30): 7.0}), ""),
        ("after_text_no_ts_yet", [1000, 2000], dict(seed=3, bumps={3000: 12.0, ids["sot"]: 20.0, ids["transcribe"]: 19.0, ids["tok_not"]: 18.0, ids["sot"] + 5 if V >= 51865 else 400: 17.0}), ""),
        ("after_opening_timestamp", [T(10)], dict(seed=4, bumps={T(40): 15.0, 4000: 9.0}), ""),
        ("after_timestamp_zero_then_text", [T(0), 1234], dict(seed=5, scale=0.5, ts_shift=7.0, bumps={T(0): 9.0, 777: 5.0}), "D2"),
        ("closing_timestamp_needs_ts_or_eot", [T(10), 1000, T(60)], dict(seed=6, bumps={5000: 15.0, T(60): 9.0, T(59): 12.0, T(61): 8.5, eot: 7.0}), ""),
        ("after_timestamp_pair_no_timestamp", [T(10), 1000, T(60), T(60)], dict(seed=7, bumps={T(100): 15.0, 6000: 8.0}), ""),
        ("text_after_pair_monotonic", [T(10), 1000, T(60), T(60), 2000], dict(seed=8, scale=0.5, ts_shift=7.0, bumps={T(60): 12.0, T(20): 14.0, T(61): 9.0, 3000: 4.0}), "D2"),
        ("timestamp_mass_mid_sequence", [T(5), 1000, 1001], dict(seed=9, scale=0.3, ts_shift=4.0, bumps={2000: 6.0}), "D2"),
        ("text_beats_timestamp_mass", [T(5), 1000, 1001], dict(seed=10, scale=0.3, ts_shift=-2.0, bumps={2000: 9.0, T(700): 5.0}), "D2"),
        ("eot_allowed_after_text", [T(0), 1000], dict(seed=11, bumps={eot: 16.0, 3000: 9.0}), "D2"),
    ]


def rules_case_logits(V: int, recipe: dict) -> np.ndarray:
    ids = gf.special_token_ids(V)
    rng = np.random.default_rng(1000 + recipe["seed"])
    lg = (rng.standard_normal(V) * recipe.get("scale", 1.0)).astype(np.float32)
    lg[ids["beg"]:] += np.float32(recipe.get("ts_shift", 0.0))
    for k, v in recipe.get("bumps", {}).items():
        lg[int(k)] = np.float32(v)
    return lg


def hf_delta_filter(V: int, tokens, logits: np.ndarray) -> None:
    """Applies to whisper.cpp-style rules what transformers' rules do in addition (D1, D2 above); in place."""
    beg = gf.special_token_ids(V)["beg"]
    n = len(tokens)
    if n == 0:
        logits[:beg] = -np.inf                                                        # D1
    ts = [t for t in tokens if t >= beg]
    if ts:                                                                            # D2
        closing = tokens[-1] >= beg and not (n < 2 or tokens[-2] >= beg)
        logits[beg:ts[-1] + (0 if closing else 1)] = -np.inf


def make_generation_config(V: int):
    from transformers import GenerationConfig
    ids = gf.special_token_ids(V)
    gc = GenerationConfig(decoder_start_token_id=ids["sot"], eos_token_id=ids["eot"], pad_token_id=ids["eot"], bos_token_id=ids["eot"],
                          suppress_tokens=suppress_list(V), begin_suppress_tokens=[220, ids["eot"]],
                          no_timestamps_token_id=ids["tok_not"], max_initial_timestamp_index=50, return_timestamps=True,
                          is_multilingual=V >= 51865, max_length=448, do_sample=False, num_beams=1)
    if V >= 51865:
        gc.lang_to_id = {"<|en|>": ids["sot"] + 1, "<|de|>": ids["sot"] + 3}
        gc.task_to_id = {"transcribe": ids["transcribe"], "translate": ids["translate"]}
    return gc


def part_a():
    import torch
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    out = {}
    for arch, V in VOCABS.items():
        ids = gf.special_token_ids(V)
        npr = n_prompt(V)
        procs = [SuppressTokensLogitsProcessor(suppress_list(V)), SuppressTokensAtBeginLogitsProcessor([220, ids["eot"]], npr),
                 WhisperTimeStampLogitsProcessor(make_generation_config(V), npr)]
        prompt = [ids["sot"]] + ([ids["sot"] + 1, ids["transcribe"]] if V >= 51865 else [])
        for name, ctx, recipe, _ in rules_cases(V):
            scores = torch.from_numpy(rules_case_logits(V, recipe))[None, :]
            inp = torch.tensor([prompt + list(ctx)], dtype=torch.long)
            for p in procs:
                scores = p(inp, scores)
            lp = torch.nn.functional.log_softmax(scores.float(), dim=-1)[0].numpy()
            fin = np.isfinite(scores[0].numpy())
            out[f"{arch}/{name}/finite"] = np.packbits(fin)
            out[f"{arch}/{name}/argmax"] = np.int32(int(lp.argmax()))
            out[f"{arch}/{name}/logprob"] = np.float32(lp.max())
            print(f"A {arch:9s} {name:36s} finite {int(fin.sum()):6d} argmax {int(lp.argmax()):6d} logprob {float(lp.max()):.4f}")
    return out


def part_b():
    """The raw greedy token stream of the first 30 s window: transformers' generic generation loop (GenerationMixin.generate,
    greedy search) driving the model with transformers' Whisper logits processors. WhisperGenerationMixin.generate is
    bypassed on purpose: it cuts the stream into segments and seeks through the padded 30 s input with rules of its own."""
    import torch
    from transformers.generation.logits_process import LogitsProcessorList, WhisperTimeStampLogitsProcessor
    from transformers.generation.utils import GenerationMixin
    from oracle.hf_golden import hf_mel, hf_model_from_ggml
    torch.set_grad_enabled(False)
    out = {}
    for arch, V in VOCABS.items():
        ids = gf.special_token_ids(V)
        path = os.path.join(tempfile.gettempdir(), f"wmi_{arch}_varied.bin")
        gf.write_synthetic_model(path, arch, seed=0, decisive=4.0, style="varied")
        hf, _ = hf_model_from_ggml(path)
        gc = make_generation_config(V)          # suppress_tokens / begin_suppress_tokens become transformers' own processors
        prompt = [ids["sot"]] + ([ids["sot"] + 1, ids["transcribe"]] if V >= 51865 else [])
        for seed, secs in SEQ_CLIPS:
            pcm = synth_pcm(seed, int(round(secs * 16000)))
            feats = torch.from_numpy(hf_mel(pcm, 128 if V == 51866 else 80)[None])
            procs = LogitsProcessorList([WhisperTimeStampLogitsProcessor(gc, len(prompt))])
            seq = GenerationMixin.generate(hf, input_features=feats, decoder_input_ids=torch.tensor([prompt]), generation_config=gc,
                                           logits_processor=procs, max_new_tokens=N_MAX, do_sample=False, num_beams=1)[0].tolist()
            assert seq[:len(prompt)] == prompt
            seq = seq[len(prompt):]
            out[f"{arch}/s{seed}"] = np.asarray(seq, dtype=np.int32)
            beg = ids["beg"]
            print(f"B {arch:9s} clip {seed} ({secs} s): {len(seq)} tokens, {sum(t >= beg for t in seq)} timestamps, {len(set(seq))} distinct, "
                  f"eot {seq.count(ids['eot'])}, head {seq[:10]}")
    return out


def part_c():
    """Language detection as transformers does it (WhisperGenerationMixin.detect_language: one decoder step on the start token,
    the logits of the language tokens only): those logits for every clip, multilingual vocabularies. whisper.cpp's
    whisper_lang_auto_detect takes the softmax over the same tokens."""
    import torch
    from oracle.hf_golden import hf_mel, hf_model_from_ggml
    torch.set_grad_enabled(False)
    out = {}
    for arch, V in VOCABS.items():
        if V < 51865:
            continue
        ids = gf.special_token_ids(V)
        n_lang = V - 51765 - 1
        path = os.path.join(tempfile.gettempdir(), f"wmi_{arch}_varied.bin")
        gf.write_synthetic_model(path, arch, seed=0, decisive=4.0, style="varied")
        hf, _ = hf_model_from_ggml(path)
        for seed, secs in SEQ_CLIPS:
            pcm = synth_pcm(seed, int(round(secs * 16000)))
            feats = torch.from_numpy(hf_mel(pcm, 128 if V == 51866 else 80)[None])
            logits = hf(input_features=feats, decoder_input_ids=torch.tensor([[ids["sot"]]])).logits[0, -1]
            lang = logits[ids["sot"] + 1: ids["sot"] + 1 + n_lang].numpy().astype(np.float32)
            out[f"{arch}/s{seed}"] = lang
            top = np.argsort(-lang)[:2]
            print(f"C {arch:9s} clip {seed}: language {int(top[0])} (logit {lang[top[0]]:.3f}), runner-up {int(top[1])} ({lang[top[1]]:.3f})")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--only-lang", action="store_true", help="regenerate the language-detection fixture only")
    args = ap.parse_args()
    c = part_c()
    if args.write:
        os.makedirs(GOLDEN, exist_ok=True)
        np.savez_compressed(os.path.join(GOLDEN, "hf_lang.npz"), **c)
    if args.only_lang:
        return
    a, b = part_a(), part_b()
    if args.write:
        np.savez_compressed(os.path.join(GOLDEN, "hf_rules.npz"), **a)
        np.savez_compressed(os.path.join(GOLDEN, "hf_sequences.npz"), **b)
        print("wrote", GOLDEN)


if __name__ == "__main__":
    main()
