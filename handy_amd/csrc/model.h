// Whisper model: legacy ggml file -> device-resident fp16/fp32 tensors in the layouts the kernels read.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "model_parse.h"

namespace wmi {

struct EncLayer {
    float *ln1_w, *ln1_b, *bqkv, *bo, *ln2_w, *ln2_b, *b1, *b2;
    f16 *wqkv, *wo, *fc1, *fc2;
};
struct DecLayer {
    float *ln1_w, *ln1_b, *bqkv, *bo, *lnx_w, *lnx_b, *bxq, *bxo, *ln2_w, *ln2_b, *b1, *b2;
    f16 *wqkv, *wo, *xq, *xo, *fc1, *fc2;
    QuantW q_wqkv, q_wo, q_xq, q_xo, q_fc1, q_fc2;      // the same matrices as the file's blocks (ttype 0: the file holds them as f16 / f32)
};

struct Model {
    HParams hp;
    Vocab vocab;
    int device = 0;
    int conv1_kpad = 0;          // 3*n_mels rounded up to the GEMM K-step
    int n_vocab_pad = 0;         // token-embedding rows rounded up to the GEMM N tile
    MelTables mel{};
    f16 *conv1_w = nullptr, *conv2_w = nullptr, *d_te = nullptr, *xkv_w = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *e_pe = nullptr, *ln_post_w = nullptr, *ln_post_b = nullptr;
    float *d_pe = nullptr, *d_ln_w = nullptr, *d_ln_b = nullptr, *xkv_b = nullptr;
    unsigned *nst_bitmap = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    std::vector<void *> allocs;  // every device allocation, for release
    size_t bytes = 0;
    std::string type_name;

    ~Model();
    VocabIds vocab_ids() const;
    int token_lang(int lang_id) const { return vocab.sot + 1 + lang_id; }
};

// Parses `data` (a whole ggml file image) and uploads it to `device`. Returns nullptr (after logging) on any
// malformed, truncated or unsupported input.
Model *load_model(const uint8_t *data, size_t size, int device);

// process-wide switch read by load_model (see model.cpp)
void set_quant_resident(bool on);
bool quant_resident();



}  // namespace wmi
