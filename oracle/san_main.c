/* Driver of the sanitizer build of the CPU checker (make -C oracle SAN=1). TEST INFRASTRUCTURE ONLY.
 * usage: oracle_san model.bin [seconds]  -- log-mel, one encoder window, a teacher-forced decode and whisper_full (greedy and
 * beam search) over a synthetic clip; prints the token count. Memory errors and undefined behaviour end it through the
 * sanitizers. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "whisper_oracle.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: oracle_san model.bin [seconds]\n"); return 2; }
    const int secs = argc > 2 ? atoi(argv[2]) : 3;
    wo_model *m = wo_load(argv[1]);
    if (!m) { printf("rejected\n"); return 3; }
    wo_state *s = wo_new_state(m);
    if (!s) return 4;
    const int n = secs * 16000;
    float *pcm = (float *)malloc(sizeof(float) * (size_t)n);
    unsigned x = 12345u;
    for (int i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        pcm[i] = 0.1f * sinf(2.0f * 3.14159265f * 440.0f * (float)i / 16000.0f) + 0.02f * ((float)(x >> 9) / 4194304.0f - 1.0f);
    }
    if (wo_pcm_to_mel(s, pcm, n) != 0) return 5;
    if (wo_encode(s, 0, 0) != 0) return 6;
    int hp[11];
    wo_hparams(m, hp);
    const int sot = wo_token_special(m, "sot");
    int32_t prompt[3] = {sot, sot + 1, wo_token_special(m, "transcribe")};
    const int np = hp[0] >= 51865 ? 3 : 1;
    if (wo_decode(s, 0, prompt, np, 0, 1) != 0) return 7;
    int total = 0;
    for (int strategy = 0; strategy < 2; strategy++) {
        wo_params p = wo_default_params(strategy);
        p.language = "en";
        p.beam_size = 2; p.best_of = 2; p.max_tokens = 24;      /* a short run: the sanitizers cost a factor of three */
        if (wo_full(s, &p, pcm, n) != 0) return 8;
        for (int i = 0; i < wo_full_n_segments(s); i++) total += wo_full_n_tokens(s, i);
    }
    printf("ok %d tokens\n", total);
    free(pcm);
    wo_free_state(s);
    wo_free_model(m);
    return 0;
}
