// The ggml file parser of the engine: host code only (no HIP), so that it also builds with gcc under
// -fsanitize=address,undefined (handy_amd/csrc/Makefile: parse_san) and can be driven over truncated and corrupt files.
// Handy hands the loader whatever is on disk under the model's name -- its downloader resumes partial files and only
// compares sizes (reference: src-tauri/src/managers/model.rs:361-405, 496-513) -- so every length and count in the file
// is untrusted.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

namespace wmi {

struct HParams {
    int n_vocab = 0, n_audio_ctx = 0, n_audio_state = 0, n_audio_head = 0, n_audio_layer = 0;
    int n_text_ctx = 0, n_text_state = 0, n_text_head = 0, n_text_layer = 0, n_mels = 0, ftype = 0;
};

struct Vocab {
    int n_vocab = 0;
    std::vector<std::string> id_to_token;
    std::unordered_map<std::string, int> token_to_id;
    int eot = 50256, sot = 50257, translate = 50357, transcribe = 50358, solm = 50359, prev = 50360, nosp = 50361, tok_not = 50362, beg = 50363;
    int space = -1;
    bool multilingual() const { return n_vocab >= 51865; }
    int num_languages() const { return n_vocab - 51765 - (multilingual() ? 1 : 0); }
};

struct TensorRef {            // a tensor inside the file image
    int ttype = 0;            // ggml type: 0 F32, 1 F16, 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0
    std::vector<int> ne;
    const uint8_t *data = nullptr;
    size_t n = 0;             // elements
};

struct ParsedFile {
    HParams hp;
    std::vector<float> filters;      // [n_mels][201]
    int n_file_vocab = 0;            // tokens the file itself holds (the rest are synthesised)
    Vocab vocab;
    std::unordered_map<std::string, TensorRef> tensors;
};

constexpr int kNumLangs = 100;
int lang_id_from_str(const char *s);
const char *lang_str(int id);
const char *lang_str_full(int id);

int ggml_block_bytes(int ttype);     // bytes per 32-weight block of a block-quantised ggml type, 0 otherwise
// Expands a tensor to f32 (block formats: 32 elements per block; low nibbles are elements 0..15, high 16..31).
void expand_f32(const TensorRef &t, float *dst);

// Header, mel filters, vocabulary and tensor directory of a whole file image. False (after logging) on anything
// malformed, truncated or unsupported; on success every TensorRef points into `data` with its bytes inside the image.
bool parse_model_file(const uint8_t *data, size_t size, ParsedFile &out);
// Every tensor the engine reads exists with the element count the hyper-parameters imply.
bool check_model_tensors(const ParsedFile &f);

// Text -> token ids as whisper.cpp's tokenize() does it (initial_prompt, whisper_tokenize): words by the GPT-2 pattern, each
// covered by the longest vocabulary entries.
std::vector<int> tokenize(const Vocab &v, const std::string &text);

}  // namespace wmi
