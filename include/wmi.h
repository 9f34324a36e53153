/*
 * wmi.h -- extensions of libwhisper_mi355.so next to the whisper.h ABI.
 *
 * Handy serialises every transcription behind one mutex, one utterance per call
 * (reference: src-tauri/src/managers/transcription.rs:39, :383); its only concurrent caller,
 * the Discord voice loop, queues N `transcribe` calls on that mutex
 * (src-tauri/src/discord_conversation.rs:470-476). The batch entry points below are what a
 * batch-aware caller (or an offline bulk driver) binds instead: the same whisper_full semantics
 * applied to B independent utterances at once on one GPU. They have no counterpart in the
 * reference; whisper_full_with_state() is wmi_full_batch() with B = 1.
 */
#ifndef WMI_H
#define WMI_H

#include "whisper.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A state sized for up to max_utts utterances in flight, each with up to max_decoders sequences
 * (beam size / best_of, <= 8), encoder work issued in chunks of enc_chunk utterances (0 = default). */
WHISPER_API struct whisper_state *wmi_init_state_batch(struct whisper_context *ctx, int max_utts, int max_decoders, int enc_chunk);

/* whisper_full semantics for n_utts utterances. samples[i] points to n_samples[i] floats
 * (16 kHz mono, [-1,1]) in HOST memory, or in DEVICE memory when samples_on_device != 0.
 * Returns 0 when the batch ran, whatever became of single utterances: an utterance that fails on its own (no language
 * detected in a clip too short for it, no room left in the K/V page pool, ...) carries its whisper_full error code in
 * wmi_batch_rc() and has no segments, and the others are unaffected -- as when Handy's callers each make their own
 * transcribe call (reference: src-tauri/src/discord_conversation.rs:470-476). Non-zero: the batch as a whole failed
 * (bad arguments, a HIP error, an abort callback). */
WHISPER_API int wmi_full_batch(struct whisper_context *ctx, struct whisper_state *state, struct whisper_full_params params,
                               const float *const *samples, const int *n_samples, int n_utts, int samples_on_device);

/* Readout per utterance (whisper_full_*_from_state reads utterance 0). */
WHISPER_API int wmi_batch_n_utts(struct whisper_state *state);
WHISPER_API int wmi_batch_n_segments(struct whisper_state *state, int utt);
WHISPER_API int64_t wmi_batch_segment_t0(struct whisper_state *state, int utt, int seg);
WHISPER_API int64_t wmi_batch_segment_t1(struct whisper_state *state, int utt, int seg);
WHISPER_API const char *wmi_batch_segment_text(struct whisper_state *state, int utt, int seg);
WHISPER_API int wmi_batch_n_tokens(struct whisper_state *state, int utt, int seg);
WHISPER_API whisper_token_data wmi_batch_token_data(struct whisper_state *state, int utt, int seg, int tok);
WHISPER_API int wmi_batch_lang_id(struct whisper_state *state, int utt);
WHISPER_API int wmi_batch_rc(struct whisper_state *state, int utt);        /* 0, or the whisper_full error code of this utterance */
WHISPER_API int wmi_batch_n_fallbacks(struct whisper_state *state, int utt);

/* Benchmark control (no upstream counterpart): when n > 0 every decoded sequence is exactly n
 * tokens long -- the end-of-text token is masked before position n and forced at it. Synthetic
 * audio and random weights have no natural transcript length; this fixes the decode workload. */
WHISPER_API void wmi_set_force_tokens(struct whisper_state *state, int n);
/* Encoder/decoder overlap inside one wmi_full_batch call: the batch is cut into n index ranges; range k decodes (HBM-bound)
 * while the encoder (MFMA-bound) is still working on ranges k+1.. on a second stream. 1 = encode everything first. Results do
 * not depend on n. */
/* Window memo for callers that transcribe a growing recording again and again (Handy's live coaching peeks at the
 * recording every 3 s and transcribes all of it: reference src-tauri/src/live_coaching.rs:84-97). With max_windows > 0 the
 * state remembers the outcome of interior 30 s windows of host-PCM calls; a later call whose samples, parameters, language,
 * text context and earlier windows are identical takes a remembered window's segments instead of encoding and decoding it.
 * The results are the ones the call would have produced without the memo. 0 turns it off and forgets everything. */
WHISPER_API void wmi_set_window_cache(struct whisper_state *state, int max_windows);
WHISPER_API void wmi_get_window_cache_stats(struct whisper_state *state, long long *hits, long long *stores);
WHISPER_API void wmi_set_pipeline_groups(struct whisper_state *state, int n);
/* 2: alternate the encoder chunks of a batch between two streams with their own workspaces (a second ~0.9 GB set for
 * large-v3-turbo at 16 utterances per chunk), so one chunk's kernels fill the CUs another chunk's draining kernels leave idle.
 * Per-kernel event times then overlap; leave it at 1 when measuring single kernels. Returns 0, or -1 when out of memory. */
WHISPER_API int wmi_set_encoder_lanes(struct whisper_state *state, int n);
/* Beam-search candidate rule: 0 = deterministic k most probable tokens per beam (default),
 * 1 = k draws from the token distribution with a per-decoder mt19937 (what upstream does). */
WHISPER_API void wmi_set_beam_candidates_sampled(struct whisper_state *state, int sampled);

/* Log-mel only, for n_utts utterances already in DEVICE memory (pcm_dev + offsets[i], n_samples[i] floats):
 * runs the frontend kernels and leaves the raw log-mel on the device. Used by the mel roofline probe. */
WHISPER_API int wmi_mel_batch_device(struct whisper_context *ctx, struct whisper_state *state, const float *pcm_dev,
                                     const long long *offsets, const int *n_samples, int n_utts);

/* The same, ending in what whisper_pcm_to_mel leaves: the clamped and scaled log-mel ((max(x, max - 8) + 4) / 4, the maximum taken
 * per utterance) of every utterance, written to the caller's device buffer mel_out_dev[n_utts][n_mel][n_frames_out] (float;
 * n_frames_out a multiple of 4; frames past an utterance's audio carry the floor value). The whole frontend of
 * whisper_pcm_to_mel_with_state for a batch already in device memory (reference call site: the engine behind
 * src-tauri/src/managers/transcription.rs:413-415 computes it per clip on the CPU). */
WHISPER_API int wmi_mel_batch_device_normalized(struct whisper_context *ctx, struct whisper_state *state, const float *pcm_dev,
                                                const long long *offsets, const int *n_samples, int n_utts, float *mel_out_dev, int n_frames_out);

/* The log-mel frontend for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc). No counterpart in the reference: it is whisper_pcm_to_mel_with_state() followed by
 * wmi_get_mel(), with the PCM (n_samples floats) and the result (normalised log-mel, [n_mel][n_frames_out] floats, frames
 * past the audio at the floor value) both in the CALLER's device memory on the context's GPU, and the kernels'
 * intermediate data in a fixed 1.5 MB device array of the library instead of buffers it would allocate. It calls no
 * hipMalloc / hipFree or other allocating routine: it enqueues one 256-byte memset and two kernels on `stream` (a
 * hipStream_t, or NULL for the null stream) and returns without waiting for them.
 * Limits: n_samples <= 480000 (30 s) and n_frames_out <= 6000; beyond them the call returns -2 and does nothing.
 * Every call on a GPU uses that GPU's one array: calls that may overlap on one GPU (two streams, two threads) race. */
WHISPER_API int wmi_pcm_to_mel_device_noalloc(struct whisper_context *ctx, const float *pcm_dev, int n_samples, float *mel_out_dev,
                                              int n_frames_out, void *stream);

/* Inspection hooks used by the parity tests (all copy to host memory). */
WHISPER_API int wmi_get_mel(struct whisper_state *state, int utt, float *out, int n_frames);          /* normalised [n_mel][n_frames] */
WHISPER_API int wmi_get_encoder_output(struct whisper_state *state, float *out);                       /* last whisper_encode: [n_ctx][n_state] */
WHISPER_API int wmi_get_cross_kv(struct whisper_state *state, int layer, float *k_out, float *v_out);  /* utterance slot 0: [n_ctx][n_state] each */
WHISPER_API int wmi_decode_logits_all(struct whisper_context *ctx, struct whisper_state *state, const whisper_token *tokens, int n_tokens, int n_past, float *logits_out);
WHISPER_API int wmi_debug_gemm(int M, int N, int K, const float *A, const float *W, float *C);          /* C = f16(A) f16(W)^T on the device */

WHISPER_API int wmi_debug_gemm_gelu(int M, int N, int K, const float *A, const float *W, const float *bias, unsigned short *out); /* f16 bits of gelu(f16(A) f16(W)^T + bias): the fc1 epilogue */
/* The loader's device-side expansion of ggml block-quantised tensors (types 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0; 32 weights a
 * block) on host buffers: blocks -> out[n_blocks * 32] as fp16 bit patterns. */
WHISPER_API int wmi_debug_expand_blocks(int ggml_type, const void *blocks, long long n_blocks, unsigned short *out);
WHISPER_API int wmi_debug_quant_rows(int ggml_type, const void *blocks, int N, int K, const unsigned short *a_f16, int rows, float *out_blocks, float *out_f16);   /* test hook: few-row GEMM on blocks vs on their fp16 expansion */
/* Test hook: the encoder's attention kernel on host buffers. q, k, v: fp16 bit patterns [B][H][T][64], q already in the kernel's
 * units (the 1/8 applied: p = exp(q.k - max)); out: fp16 bit patterns [B][T][H*64]. */
WHISPER_API int wmi_debug_enc_attention(int B, int H, int T, const unsigned short *q, const unsigned short *k, const unsigned short *v, unsigned short *out);

#ifdef WMI_HARNESS
/* Benchmark harness (libwhisper_mi355_harness.so, built from the same sources with -DWMI_HARNESS; the product library does not
 * carry these): tile-shape / schedule variants of the GEMM, cycle stamps, LayerNorm and LDS access-pattern A/B kernels. */
WHISPER_API int wmi_debug_read_gemm_stamps(unsigned long long *out, int n);                             /* harness: per-section cycle sums of variant 14 */
WHISPER_API void wmi_debug_set_tile_group(int rows);                                                    /* harness knob: tile rows per L2 group */
WHISPER_API int wmi_debug_gemm_variant_diff(int M, int N, int K, int variant, int ref_variant, float *max_abs_diff); /* harness */
WHISPER_API int wmi_bench_lds_pattern(int pattern, int do_read, int iters, float *ms_out);                /* 16-byte LDS access patterns, one workgroup per CU (harness) */
WHISPER_API int wmi_bench_layer_norm(int M, int d, int variant, int iters, float *ms_out);            /* LayerNorm A/B harness (variant 100 = the shipped kernel) */
WHISPER_API int wmi_bench_gemm(int M, int N, int K, int variant, int iters, float *ms_out);            /* tile-shape A/B harness, random fp16 operands */
WHISPER_API void wmi_debug_set_enc_attention(int variant);                                              /* harness knob: 0 enc_attention_kernel, 1 enc_attention_pipe_kernel */
WHISPER_API int wmi_bench_enc_attention(int B, int H, int T, int iters, float *ms_out);                  /* encoder attention alone, random fp16 data (harness) */
#endif

/* Per-kernel-class device time, measured with HIP events on the engine's stream. */
enum wmi_kernel_class { WMI_K_MEL = 0, WMI_K_ENC_GEMM, WMI_K_ENC_ATTN, WMI_K_ENC_OTHER, WMI_K_DEC_GEMM, WMI_K_DEC_CROSS_ATTN, WMI_K_DEC_OTHER, WMI_K_COUNT };
typedef struct wmi_kernel_stat { double ms; double work; long long launches; } wmi_kernel_stat;  /* work: flop (GEMM/attention) or bytes */
WHISPER_API void wmi_set_profiling(struct whisper_state *state, int enabled);
WHISPER_API void wmi_set_profiling_classes(struct whisper_state *state, unsigned class_mask);   /* bit k = class k; 0 turns profiling off */
WHISPER_API int wmi_get_kernel_stats(struct whisper_state *state, wmi_kernel_stat *out /* [WMI_K_COUNT] */);
WHISPER_API void wmi_reset_kernel_stats(struct whisper_state *state);

/* Host-side clocks accumulated by the state, ms: [0] log-mel phase, [1] encoder launches, [2] decoder passes,
 * [3] of which building + launching, [4] of which waiting for the stream; [5] number of decoder passes. */
/* Block-quantised model files (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0: the catalogue formats of
 * /root/reference/src-tauri/src/managers/model.rs:97,135 among them) can keep their decoder weight matrices resident as the file's
 * own blocks next to the fp16 copies; decode passes of at most 16 rows (one clip, its beams) then read the blocks and expand them in
 * registers to the fp16 values the load-time expansion writes -- same results bit for bit, a third to a half of the weight bytes
 * per token. wmi_set_quant_resident(1) before whisper_init_* (or WMI_QUANT_RESIDENT=1) turns the resident copies on for models
 * loaded afterwards; it is off by default because those passes are bound by launch latency, not bytes, and the expansion costs
 * more than it saves today. wmi_set_quant_decode(state, 0) makes a state read the fp16 copies although blocks are resident. */
/* wmi_set_decoder_lanes(state, 2): decoder passes of 32 rows or more run as two halves of their rows on two HIP streams (a row's
 * results do not depend on which rows share its launches, so the outcome is the same to the bit). Off by default (1 lane): the
 * half-size launches take as long as the full-size ones (they are bound by their own dependent chains) but did not overlap on
 * the boxes this was measured on. */
WHISPER_API int wmi_set_decoder_lanes(struct whisper_state *state, int n);
/* Decoder passes whose shape (rows, sequences, groups, audio context, ...) repeats are replayed as one HIP graph launch instead of
 * the ten launches per layer they consist of; what differs between such passes travels in a metadata buffer the kernels read.
 * Results are the same either way. On by default (WMI_PASS_GRAPHS=0 in the environment or on = 0 here turns it off). */
WHISPER_API void wmi_set_pass_graphs(struct whisper_state *state, int on);
WHISPER_API void wmi_set_quant_resident(int on);
WHISPER_API int wmi_set_quant_decode(struct whisper_state *state, int on);
WHISPER_API int wmi_get_host_timings(struct whisper_state *state, double *out /* [6] */);

/* ---- The step in front of the path, for bulk files: resample to 16 kHz frames + voice-activity gate ----------------------
 * Handy's recorder does this per 30 ms of microphone audio on the CPU; an offline driver with hours of 44.1 / 48 kHz
 * files in device memory calls these instead and hands the result to wmi_full_batch(..., samples_on_device = 1). The speech
 * probability per frame (Silero, run by the vad-rs crate in the reference) stays with the caller. All device pointers are on the
 * current HIP device; `stream` is a hipStream_t or NULL. */

/* FrameResampler::new(in_hz, 16000, 30 ms) + push(all samples) + finish()
 * (reference: src-tauri/src/audio_toolkit/audio/resampler.rs:16-98; the arithmetic is rubato 0.16.2's FftFixedIn<f32> with a
 * 1024-sample chunk, resampler.rs:5,24). wmi_ingest_n_frames: how many 480-sample frames n_in samples at in_hz give, -1 for bad
 * arguments. wmi_ingest_resample: writes those frames (the last one zero-padded) to frames_dev; enqueues on `stream` and
 * returns. 0 ok, -1 bad arguments, -2 n_frames_cap too small, -3 rate not supported (the transform sizes must factor into
 * 2, 3, 5, 7 and fit the LDS: every rate from 8 to 48 kHz in common use does, 96 kHz does not), -4 HIP error. */
WHISPER_API long long wmi_ingest_n_frames(int in_hz, long long n_in);
WHISPER_API int wmi_ingest_resample(const float *pcm_dev, long long n_in, int in_hz, float *frames_dev, long long n_frames_cap, void *stream);

/* SmoothedVad::push_frame over a whole recording, then the recorder's handle_frame
 * (reference: src-tauri/src/audio_toolkit/vad/smoothed.rs:41-96, audio_toolkit/audio/recorder.rs:290-295;
 * managers/audio.rs:121-123 uses prefill 15, hangover 15, onset 2 on `probability > 0.3`). Host only. is_voice[i] != 0: the
 * detector called frame i speech. Returns the number of frames appended to the output and writes the source frame of each, in
 * order, to src_out (up to cap; a frame still in the pre-roll buffer when speech starts again is appended again, as in the
 * reference). -1 for bad arguments. */
WHISPER_API long long wmi_ingest_gate_plan(const unsigned char *is_voice, long long n_frames, int prefill, int hangover, int onset,
                                           long long *src_out, long long cap);

/* The gather: out_dev = the frames src[0..n_src) of frames_dev back to back, and, when that is less than one second but not
 * empty, zeros up to 1.25 s (reference: src-tauri/src/managers/audio.rs:402-411). src is in host memory; both device pointers
 * 16-byte aligned. *n_out = samples written. Waits for `stream`. 0 ok, -1 bad arguments, -2 out_cap too small (*n_out says what
 * is needed), -4 HIP error. */
WHISPER_API int wmi_ingest_gather(const float *frames_dev, long long n_frames, const long long *src, long long n_src, float *out_dev,
                                  long long out_cap, long long *n_out, void *stream);

/* ---- Device memory of the HIP runtime THIS library is linked against ---------------------------------------------------------
 * For callers (ctypes, Rust) that have no HIP binding of their own and must not guess which copy of the runtime the library uses:
 * a second runtime in one process (another ROCm major's soname, the copy a Python wheel carries) does not share its allocations.
 * All act on the calling thread's current device. No counterpart in the reference (its callers hold host vectors). */
WHISPER_API void *wmi_dev_alloc(size_t bytes);                                     /* hipMalloc; NULL on failure */
WHISPER_API void wmi_dev_free(void *p);
WHISPER_API int wmi_dev_upload(void *dst_dev, const void *src_host, size_t bytes); /* hipMemcpy (blocking); 0 ok */
WHISPER_API int wmi_dev_download(void *dst_host, const void *src_dev, size_t bytes);
WHISPER_API int wmi_dev_memset(void *dst_dev, int byte, size_t bytes);
WHISPER_API int wmi_dev_sync(void);                                                /* hipDeviceSynchronize */
WHISPER_API long long wmi_dev_free_bytes(void);                                    /* hipMemGetInfo: free device memory, -1 on failure */
WHISPER_API void *wmi_dev_stream_create(void);                                     /* a non-blocking hipStream_t; NULL on failure */
WHISPER_API int wmi_dev_stream_sync(void *stream);
WHISPER_API void wmi_dev_stream_destroy(void *stream);

/* ---- Node level: one engine per GPU, data-parallel by utterance -----------------------------------------------------------
 * No counterpart in the reference (one engine behind one mutex: src-tauri/src/managers/transcription.rs:39, :383; the only
 * concurrent caller queues on it: src-tauri/src/discord_conversation.rs:470-476). What a bulk driver binds: one process, one
 * worker thread and one engine (full weights replica) per GPU; the workers pull batches of `batch` utterances from an atomic
 * counter, so a GPU whose utterances decode for longer pulls fewer; results are fixed-size records in utterance order. With
 * more than one distinct GPU the records travel to the first device over RCCL point-to-point (librccl is loaded at run time;
 * WMI_NODE_GATHER=host, a missing library or a refused device list leave them in host memory, where the workers also put
 * them) -- the only exchange between GPUs; there is no tensor parallelism. */
WHISPER_API struct whisper_context *wmi_ctx_for_device(const char *path_model, int device);   /* whisper_init_from_file_with_params_no_state on GPU `device` */

#define WMI_NODE_MAX_TOKENS 240
typedef struct wmi_node_record {
    int utt;            /* index of the utterance in the job */
    int rc;             /* 0, or this utterance's whisper_full error code (wmi_batch_rc), or its batch's when the whole batch failed */
    int device;         /* which of the node's engines ran it (index into the device list) */
    int lang_id;
    int n_segments;
    int n_tokens;       /* tokens of all segments; the first WMI_NODE_MAX_TOKENS are kept (a 30 s window yields at most 224) */
    int64_t t0, t1;     /* start of the first segment, end of the last (centiseconds) */
    whisper_token tokens[WMI_NODE_MAX_TOKENS];
} wmi_node_record;
typedef struct wmi_node wmi_node;

/* devices == NULL or n_devices == 0: every visible GPU. Engines are created as wmi_init_state_batch(ctx, batch, max_decoders,
 * enc_chunk). NULL when any engine fails to start. */
WHISPER_API wmi_node *wmi_node_init(const char *path_model, const int *devices, int n_devices, int batch, int max_decoders, int enc_chunk);
WHISPER_API void wmi_node_free(wmi_node *node);
WHISPER_API int wmi_node_n_devices(wmi_node *node);
WHISPER_API struct whisper_state *wmi_node_state(wmi_node *node, int i);       /* engine i's state, e.g. for wmi_set_force_tokens */
WHISPER_API struct whisper_context *wmi_node_context(wmi_node *node, int i);
/* whisper_full semantics for n_utts utterances in HOST memory, spread over the node's engines; records_out[u] describes
 * utterance u. Callbacks in `params` are called from the worker threads. 0 when the job ran (single utterances and single
 * batches carry their own codes in the records), -1 bad arguments, -2 a worker died. */
WHISPER_API int wmi_node_run(wmi_node *node, struct whisper_full_params params, const float *const *samples, const int *n_samples,
                             int n_utts, wmi_node_record *records_out);
WHISPER_API int wmi_node_last_gather(wmi_node *node);            /* how the last job's records were gathered: 0 host memory, 1 RCCL */
WHISPER_API int wmi_node_last_batches(wmi_node *node, int i);    /* batches engine i pulled in the last job */
/* The same queue with the caller's function in place of the engine call (tests; other engines): fn(user, worker, first, count,
 * out) fills out[0..count) for utterances first..first+count-1; `utt` and `device` are set by the queue; a non-zero return
 * marks the whole batch failed with that code. batches_per_worker (may be NULL) receives how many batches each worker pulled. */
typedef int (*wmi_node_batch_fn)(void *user, int worker, int first, int count, wmi_node_record *out);
WHISPER_API int wmi_node_run_custom(int n_workers, int n_utts, int batch, wmi_node_batch_fn fn, void *user, wmi_node_record *records_out,
                                    int *batches_per_worker);

WHISPER_API const char *wmi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WMI_H */
