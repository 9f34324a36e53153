"""Cross-check the CPU oracle against Hugging Face transformers' Whisper on seeded synthetic
weights, and (re)generate the small golden fixtures under tests/golden/.

Runs only in the build container (transformers + torch CPU); the fixtures it writes are plain
.npz data: inputs are regenerated from seeds, expected outputs are stored subsampled.
Usage: HF_HUB_OFFLINE=1 python oracle/hf_golden.py [--write]
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
from oracle import oracle as wo                # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def hf_model_from_ggml(path: str):
    import torch
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    g = gf.read_model(path)
    hp = g.hp
    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state,
                        encoder_layers=hp.n_audio_layer, decoder_layers=hp.n_text_layer,
                        encoder_attention_heads=hp.n_audio_head, decoder_attention_heads=hp.n_text_head,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state,
                        activation_function="gelu_pytorch_tanh", max_source_positions=hp.n_audio_ctx,
                        max_target_positions=hp.n_text_ctx, attn_implementation="eager")
    m = WhisperForConditionalGeneration(cfg).eval()
    sd = {}
    T = lambda n: torch.from_numpy(np.ascontiguousarray(g.tensors[n]))
    sd["model.encoder.conv1.weight"] = T("encoder.conv1.weight")
    sd["model.encoder.conv1.bias"] = T("encoder.conv1.bias").reshape(-1)
    sd["model.encoder.conv2.weight"] = T("encoder.conv2.weight")
    sd["model.encoder.conv2.bias"] = T("encoder.conv2.bias").reshape(-1)
    sd["model.encoder.embed_positions.weight"] = T("encoder.positional_embedding")
    sd["model.encoder.layer_norm.weight"] = T("encoder.ln_post.weight")
    sd["model.encoder.layer_norm.bias"] = T("encoder.ln_post.bias")
    sd["model.decoder.embed_tokens.weight"] = T("decoder.token_embedding.weight")
    sd["model.decoder.embed_positions.weight"] = T("decoder.positional_embedding")
    sd["model.decoder.layer_norm.weight"] = T("decoder.ln.weight")
    sd["model.decoder.layer_norm.bias"] = T("decoder.ln.bias")
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    amap = {"query": "q_proj", "key": "k_proj", "value": "v_proj", "out": "out_proj"}
    for side, nl in (("encoder", hp.n_audio_layer), ("decoder", hp.n_text_layer)):
        for i in range(nl):
            src, dst = f"{side}.blocks.{i}.", f"model.{side}.layers.{i}."
            for wb in ("weight", "bias"):
                sd[dst + "self_attn_layer_norm." + wb] = T(src + "attn_ln." + wb)
                sd[dst + "final_layer_norm." + wb] = T(src + "mlp_ln." + wb)
                sd[dst + "fc1." + wb] = T(src + "mlp.0." + wb)
                sd[dst + "fc2." + wb] = T(src + "mlp.2." + wb)
                if side == "decoder":
                    sd[dst + "encoder_attn_layer_norm." + wb] = T(src + "cross_attn_ln." + wb)
            for a, b in amap.items():
                sd[dst + f"self_attn.{b}.weight"] = T(src + f"attn.{a}.weight")
                if a != "key":
                    sd[dst + f"self_attn.{b}.bias"] = T(src + f"attn.{a}.bias")
                if side == "decoder":
                    sd[dst + f"encoder_attn.{b}.weight"] = T(src + f"cross_attn.{a}.weight")
                    if a != "key":
                        sd[dst + f"encoder_attn.{b}.bias"] = T(src + f"cross_attn.{a}.bias")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "k_proj.bias" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m, g


def hf_mel(pcm: np.ndarray, n_mels: int) -> np.ndarray:
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=n_mels)
    return fe(pcm, sampling_rate=16000, return_tensors="np").input_features[0]      # [n_mels, 3000]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    import torch
    torch.set_grad_enabled(False)
    wo.build()
    report = {}
    fixtures = {}
    # ---- G1: log-mel, HF feature extractor vs oracle -------------------------------------
    for n_mels, arch in ((80, "micro.en"), (128, "micro-v3")):
        path = os.path.join(tempfile.gettempdir(), f"wmi_{arch}_s0.bin")
        gf.write_synthetic_model(path, arch, seed=0)
        om = wo.OracleModel(path)
        st = wo.OracleState(om)
        for seed, secs in ((0, 1.0), (1, 5.0), (2, 29.9), (3, 30.0)):
            n = int(round(secs * 16000))
            pcm = synth_pcm(seed, n)
            mel_o = st.pcm_to_mel(pcm)
            mel_h = hf_mel(pcm, n_mels)
            nfr = 3000 if secs < 30.0 else 2999      # exactly-30 s clips: HF reflects the right edge, upstream zero-fills
            err = float(np.abs(mel_o[:, :nfr] - mel_h[:, :nfr]).max())
            report[f"mel{n_mels}_s{seed}"] = err
            fixtures[f"mel{n_mels}_s{seed}"] = mel_h[:, :nfr:8].astype(np.float32)
        st.close(); om.close()
    # ---- G2: encoder / decoder activations, HF model vs oracle ---------------------------
    for arch in ("micro.en", "micro", "micro-v3"):
        path = os.path.join(tempfile.gettempdir(), f"wmi_{arch}_s0.bin")
        gf.write_synthetic_model(path, arch, seed=0)
        hf, g = hf_model_from_ggml(path)
        om = wo.OracleModel(path)
        st = wo.OracleState(om)
        pcm = synth_pcm(7, 20 * 16000)
        mel_o = st.pcm_to_mel(pcm)
        feats = torch.from_numpy(mel_o[None, :, :3000].copy())
        enc_h = hf.model.encoder(feats).last_hidden_state[0].numpy()
        enc_o = st.encode(0)
        report[f"{arch}_enc_abs"] = float(np.abs(enc_h - enc_o).max())
        sot = om.special("sot")
        rng = np.random.default_rng(5)
        toks = [sot] + [int(t) for t in rng.integers(0, 50000, size=7)]
        out = hf(input_features=feats, decoder_input_ids=torch.tensor([toks]))
        logits_h = out.logits[0].numpy()
        logits_o = st.decode(toks, 0, 0, logits_all=True)
        report[f"{arch}_logits_abs"] = float(np.abs(logits_h - logits_o).max())
        report[f"{arch}_logits_std"] = float(logits_h.std())
        report[f"{arch}_argmax_match"] = float((logits_h.argmax(-1) == logits_o.argmax(-1)).mean())
        # incremental decode (KV cache) must equal the batched prompt pass
        st.decode(toks[:4], 0, 1)
        inc = [st.decode([toks[i]], i, 1)[0] for i in range(4, 8)]
        report[f"{arch}_kv_incremental_abs"] = float(np.abs(np.stack(inc) - logits_o[4:8]).max())
        fixtures[f"{arch}_enc"] = enc_h[::50].astype(np.float32)
        fixtures[f"{arch}_tokens"] = np.asarray(toks, dtype=np.int32)
        fixtures[f"{arch}_logits_sub"] = logits_h[:, ::97].astype(np.float32)
        fixtures[f"{arch}_logits_argmax"] = logits_h.argmax(-1).astype(np.int32)
        st.close(); om.close()
    for k, v in report.items():
        print(f"{k:32s} {v:.3e}")
    if args.write:
        os.makedirs(GOLDEN, exist_ok=True)
        np.savez_compressed(os.path.join(GOLDEN, "hf_golden.npz"), **fixtures)
        with open(os.path.join(GOLDEN, "hf_golden_report.txt"), "w") as f:
            f.write("# max abs error, CPU oracle vs Hugging Face transformers (see oracle/hf_golden.py)\n")
            for k, v in report.items():
                f.write(f"{k} {v:.6e}\n")
        print("wrote", GOLDEN)


if __name__ == "__main__":
    main()
