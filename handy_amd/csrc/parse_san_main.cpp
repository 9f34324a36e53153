// Driver of the sanitizer build of the file parser (make parse_san): reads a valid model file once, then one mutation per
// line of stdin, applies it to a fresh copy of the image and runs the parser over it:
//   valid                 the image as it is
//   trunc N               only the first N bytes
//   set32 OFFSET VALUE    a little-endian 32-bit word replaced
//   flip OFFSET BIT       one bit flipped
//   tok HEX               tokenises the text whose bytes are HEX with the (unmodified) file's vocabulary: "tok HEX -> count id id ..."
// Prints "<mutation> -> ok <tensors>" or "-> rejected" per line; a memory error or undefined behaviour ends the process
// through the sanitizers instead.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "model_parse.h"

namespace wmi {
void log_msg(int, const char *, ...) {}
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: parse_san model.bin < mutations\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary | std::ios::ate);
    if (!f) return 2;
    std::vector<unsigned char> img((size_t)f.tellg());
    f.seekg(0);
    f.read((char *)img.data(), (std::streamsize)img.size());
    char line[4096];
    while (fgets(line, sizeof line, stdin)) {
        line[strcspn(line, "\n")] = 0;
        char op[16] = {0};
        long long a = 0, b = 0;
        if (sscanf(line, "%15s %lld %lld", op, &a, &b) < 1) continue;
        if (!strcmp(op, "tok")) {
            wmi::ParsedFile F;
            if (!wmi::parse_model_file(img.data(), img.size(), F)) { printf("%s -> rejected\n", line); continue; }
            std::string text;
            const char *hex = strchr(line, ' ');
            for (hex = hex ? hex + 1 : ""; hex[0] && hex[1]; hex += 2) { unsigned v = 0; sscanf(hex, "%2x", &v); text.push_back((char)v); }
            const std::vector<int> ids = wmi::tokenize(F.vocab, text);
            printf("%s -> %zu", line, ids.size());
            for (int id : ids) printf(" %d", id);
            printf("\n");
            continue;
        }
        size_t n = img.size();
        if (!strcmp(op, "trunc")) n = (size_t)(a < 0 ? 0 : ((size_t)a > img.size() ? img.size() : (size_t)a));
        // an exact-size heap copy: reading one byte past the end of the (possibly truncated) image is an ASan report
        unsigned char *buf = (unsigned char *)malloc(n ? n : 1);
        memcpy(buf, img.data(), n);
        if (!strcmp(op, "set32") && a >= 0 && (size_t)a + 4 <= n) { const unsigned v = (unsigned)b; memcpy(buf + a, &v, 4); }
        if (!strcmp(op, "flip") && a >= 0 && (size_t)a < n) buf[a] ^= (unsigned char)(1u << (b & 7));
        {
            wmi::ParsedFile F;
            const bool ok = wmi::parse_model_file(buf, n, F) && wmi::check_model_tensors(F);
            if (ok) {            // touch what the loader would read: every tensor's bytes, expanded where it is small
                unsigned long long sum = 0;
                for (auto &kv : F.tensors) {
                    const wmi::TensorRef &t = kv.second;
                    const size_t nbytes = t.ttype == 0 ? t.n * 4 : t.ttype == 1 ? t.n * 2 : t.n / 32 * (size_t)wmi::ggml_block_bytes(t.ttype);
                    sum += t.data[0] + t.data[nbytes - 1];
                    if (t.n <= 1 << 16) { std::vector<float> v(t.n); wmi::expand_f32(t, v.data()); sum += (unsigned long long)v[t.n - 1]; }
                }
                printf("%s -> ok %zu %llu\n", line, F.tensors.size(), sum & 1);
            } else printf("%s -> rejected\n", line);
        }
        free(buf);
    }
    return 0;
}
