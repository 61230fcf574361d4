// Byte-level BPE of clip.tokenize, native (host-only C++): the tokenizer the reference calls on every CustomTextEncoder.forward
// (models/clip_encoders.py:60 -- once per training BATCH) and in utils/clip_pseudolabels.py:25.  Algorithm of the published
// openai/CLIP simple_tokenizer: a pre-token's UTF-8 bytes are the initial symbols (the last one carries the end-of-word
// mark), the adjacent pair with the lowest rank in the merges table is merged everywhere, repeatedly, until no ranked pair
// is left.  Symbols are vocabulary ids from the start -- byte b -> its position in the bytes_to_unicode order, + 256 for the
// end-of-word form, 512 + rank for merge products -- so encoding never builds strings.  Results are cached per word.
#include <stdint.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_common.h"

namespace {
struct PairHash {
    size_t operator()(uint64_t k) const { return (size_t)(k * 0x9E3779B97F4A7C15ull >> 17); }
};
}  // namespace

struct grip_bpe {
    int byte_id[256];                                               // byte value -> vocabulary id (without the end-of-word mark)
    std::unordered_map<uint64_t, int32_t, PairHash> rank;           // (id_a << 32 | id_b) -> merge rank
    std::unordered_map<std::string, std::vector<int32_t>> cache;    // pre-token bytes -> ids
    std::mutex cache_lock;                                          // ctypes releases the GIL during a call: two Python threads may share one handle
    int32_t n_merges = 0, sot = 0, eot = 0;
};

static void init_byte_ids(grip_bpe* t, std::vector<std::string>& unicode_of) {
    // bytes_to_unicode(): the printable ranges keep their code point and come first, the other bytes follow as 256 + n
    std::vector<int> bs;
    for (int b = '!'; b <= '~'; ++b) bs.push_back(b);
    for (int b = 0xA1; b <= 0xAC; ++b) bs.push_back(b);
    for (int b = 0xAE; b <= 0xFF; ++b) bs.push_back(b);
    std::vector<int> cs(bs);
    int n = 0;
    for (int b = 0; b < 256; ++b) {
        bool found = false;
        for (int x : bs) if (x == b) { found = true; break; }
        if (!found) { bs.push_back(b); cs.push_back(256 + n); ++n; }
    }
    unicode_of.assign(256, std::string());
    for (int i = 0; i < 256; ++i) {
        t->byte_id[bs[(size_t)i]] = i;
        const int c = cs[(size_t)i];          // code point < 0x800: one or two UTF-8 bytes
        std::string u;
        if (c < 0x80) u.push_back((char)c);
        else { u.push_back((char)(0xC0 | (c >> 6))); u.push_back((char)(0x80 | (c & 0x3F))); }
        unicode_of[(size_t)i] = u;
    }
}

extern "C" int grip_bpe_create(const char* merges, size_t n_bytes, grip_bpe** out) {
    GRIP_REQUIRE(merges && out, "bpe_create: null pointer");
    try {
        std::unique_ptr<grip_bpe> owner(new grip_bpe());     // released into *out only on success: no leak on a bad table or bad_alloc
        grip_bpe* t = owner.get();
        std::vector<std::string> uni;
        init_byte_ids(t, uni);
        std::unordered_map<std::string, int32_t> vocab;               // symbol string (byte-unicode alphabet) -> id
        for (int i = 0; i < 256; ++i) { vocab[uni[(size_t)i]] = i; vocab[uni[(size_t)i] + "</w>"] = 256 + i; }
        size_t pos = 0;
        while (pos < n_bytes) {
            size_t end = pos;
            while (end < n_bytes && merges[end] != '\n') ++end;
            const std::string line(merges + pos, end - pos);
            pos = end + 1;
            const size_t sp = line.find(' ');
            if (sp == std::string::npos || sp == 0 || sp + 1 >= line.size() || line.find(' ', sp + 1) != std::string::npos) continue;
            const std::string a = line.substr(0, sp), b = line.substr(sp + 1);
            auto ia = vocab.find(a), ib = vocab.find(b);
            GRIP_REQUIRE(ia != vocab.end() && ib != vocab.end(), "bpe_create: merge %d uses an unknown symbol", t->n_merges);
            t->rank[((uint64_t)(uint32_t)ia->second << 32) | (uint32_t)ib->second] = t->n_merges;
            vocab[a + b] = 512 + t->n_merges;
            ++t->n_merges;
        }
        t->sot = 512 + t->n_merges;
        t->eot = t->sot + 1;
        *out = owner.release();
        return GRIP_OK;
    } catch (...) { grip_set_error("bpe_create: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_bpe_destroy(grip_bpe* t) {
    delete t;
    return GRIP_OK;
}

extern "C" int grip_bpe_special_ids(const grip_bpe* t, int32_t* sot, int32_t* eot, int32_t* vocab_size) {
    GRIP_REQUIRE(t && sot && eot && vocab_size, "bpe_special_ids: null pointer");
    *sot = t->sot; *eot = t->eot; *vocab_size = t->eot + 1;
    return GRIP_OK;
}

// Returns a COPY: the cache may be cleared (or rehashed) by another thread the moment the lock is dropped.
static std::vector<int32_t> encode_word(grip_bpe* t, const uint8_t* w, int n) {
    const std::string key((const char*)w, (size_t)n);
    {
        std::lock_guard<std::mutex> g(t->cache_lock);
        auto hit = t->cache.find(key);
        if (hit != t->cache.end()) return hit->second;
    }
    std::vector<int32_t> sym((size_t)n);
    for (int i = 0; i < n; ++i) sym[(size_t)i] = t->byte_id[w[i]];
    sym[(size_t)n - 1] += 256;                                  // end-of-word form of the last byte
    while (sym.size() > 1) {
        int32_t best = INT32_MAX;
        uint64_t best_key = 0;
        for (size_t i = 0; i + 1 < sym.size(); ++i) {
            const uint64_t k = ((uint64_t)(uint32_t)sym[i] << 32) | (uint32_t)sym[i + 1];
            auto r = t->rank.find(k);
            if (r != t->rank.end() && r->second < best) { best = r->second; best_key = k; }
        }
        if (best == INT32_MAX) break;
        const int32_t a = (int32_t)(best_key >> 32), b = (int32_t)(best_key & 0xFFFFFFFFu);
        std::vector<int32_t> next;
        next.reserve(sym.size());
        for (size_t i = 0; i < sym.size();) {
            if (i + 1 < sym.size() && sym[i] == a && sym[i + 1] == b) { next.push_back(512 + best); i += 2; }
            else { next.push_back(sym[i]); i += 1; }
        }
        sym.swap(next);
    }
    std::lock_guard<std::mutex> g(t->cache_lock);
    if (t->cache.size() > (1u << 20)) t->cache.clear();
    return t->cache.emplace(key, std::move(sym)).first->second;
}

// One pre-token (the raw UTF-8 bytes of one match of the CLIP pre-tokenisation pattern) -> ids.
extern "C" int grip_bpe_encode_word(grip_bpe* t, const uint8_t* word, int n, int32_t* ids, int cap, int* n_out) {
    GRIP_REQUIRE(t && word && ids && n_out && n > 0, "bpe_encode_word: bad arguments");
    try {
        const std::vector<int32_t> v = encode_word(t, word, n);
        GRIP_REQUIRE((int)v.size() <= cap, "bpe_encode_word: output buffer too small");
        memcpy(ids, v.data(), v.size() * sizeof(int32_t));
        *n_out = (int)v.size();
        return GRIP_OK;
    } catch (...) { grip_set_error("bpe_encode_word: exception"); return GRIP_ERR_ARG; }
}

// A whole ASCII text that has already been cleaned and lower-cased by the host (html.unescape, whitespace collapse, .lower()):
// pre-tokenisation by the CLIP pattern restricted to ASCII
//   <|startoftext|> | <|endoftext|> | 's|'t|'re|'ve|'m|'ll|'d | [a-z]+ | [0-9] | [^\s a-z 0-9]+
// (first alternative that matches at a position wins, whitespace separates), then BPE per pre-token.  Non-ASCII text goes
// through the host's Unicode-aware pattern and grip_bpe_encode_word.
extern "C" int grip_bpe_encode_ascii(grip_bpe* t, const char* text, int n, int32_t* ids, int cap, int* n_out) {
    GRIP_REQUIRE(t && text && ids && n_out && n >= 0, "bpe_encode_ascii: bad arguments");
    try {
        auto is_space = [](unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); };   // Python's \s on ASCII
        auto is_alpha = [](unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
        auto is_digit = [](unsigned char c) { return c >= '0' && c <= '9'; };
        static const char* const contractions[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
        int count = 0, i = 0;
        auto emit = [&](const std::vector<int32_t>& v) {
            if (count + (int)v.size() > cap) return false;
            memcpy(ids + count, v.data(), v.size() * sizeof(int32_t));
            count += (int)v.size();
            return true;
        };
        while (i < n) {
            const unsigned char c = (unsigned char)text[i];
            GRIP_REQUIRE(c < 0x80, "bpe_encode_ascii: non-ASCII byte at %d", i);
            if (is_space(c)) { ++i; continue; }
            if (n - i >= 15 && !memcmp(text + i, "<|startoftext|>", 15)) { GRIP_REQUIRE(count < cap, "bpe_encode_ascii: output buffer too small"); ids[count++] = t->sot; i += 15; continue; }
            if (n - i >= 13 && !memcmp(text + i, "<|endoftext|>", 13)) { GRIP_REQUIRE(count < cap, "bpe_encode_ascii: output buffer too small"); ids[count++] = t->eot; i += 13; continue; }
            int len = 0;
            if (c == '\'')
                for (const char* k : contractions) {
                    const int l = (int)strlen(k);
                    if (n - i >= l && !memcmp(text + i, k, (size_t)l)) { len = l; break; }
                }
            if (!len) {
                if (is_alpha(c)) { while (i + len < n && is_alpha((unsigned char)text[i + len])) ++len; }
                else if (is_digit(c)) len = 1;
                else { while (i + len < n) { const unsigned char d = (unsigned char)text[i + len]; if (d >= 0x80 || is_space(d) || is_alpha(d) || is_digit(d)) break; ++len; } }
            }
            GRIP_REQUIRE(emit(encode_word(t, (const uint8_t*)text + i, len)), "bpe_encode_ascii: output buffer too small");
            i += len;
        }
        *n_out = count;
        return GRIP_OK;
    } catch (...) { grip_set_error("bpe_encode_ascii: exception"); return GRIP_ERR_ARG; }
}
