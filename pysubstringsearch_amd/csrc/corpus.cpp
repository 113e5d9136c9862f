// corpus.cpp -- deterministic synthetic corpora for bench.py and the tests
// (generator spec: SURVEY.md section 8(d)).  Integer-only xorshift64, so any
// re-implementation (tests carry a Python one) agrees bit for bit.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pss.h"

namespace {

struct Xs64 {
    uint64_t s;
    uint64_t step()
    {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    }
    uint32_t nx() { return (uint32_t)(step() >> 32); }
};

constexpr uint64_t kSeed = 88172645463325252ULL;

void gen_lines(uint8_t *out, uint64_t n, uint64_t chunk)
{
    static const char ALPHA[] = "abcdefghijklmnopqrstuvwxyz0123456789 .";
    Xs64 g{kSeed + chunk};
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t r = g.nx();
        out[i] = (r % 40 == 0) ? '\n' : (uint8_t)ALPHA[(r >> 8) % 38];
    }
}

void gen_words(uint8_t *out, uint64_t n, uint64_t chunk)
{
    constexpr uint32_t V = 65536;
    std::vector<uint8_t> letters;
    std::vector<uint32_t> start(V + 1);
    Xs64 v{0x2545F4914F6CDD1DULL};
    letters.reserve((size_t)V * 8);
    for (uint32_t w = 0; w < V; ++w) {
        start[w] = (uint32_t)letters.size();
        const uint32_t len = 3 + v.nx() % 8;
        for (uint32_t k = 0; k < len; ++k) letters.push_back((uint8_t)('a' + v.nx() % 26));
    }
    start[V] = (uint32_t)letters.size();
    Xs64 g{kSeed + chunk};
    uint64_t o = 0;
    while (o < n) {
        const uint32_t k = 1 + g.nx() % 12;
        for (uint32_t j = 0; j < k && o < n; ++j) {
            const uint32_t a = g.nx() % V;
            const uint32_t sh = g.nx() % 16;
            const uint32_t w = a >> sh;
            if (j) out[o++] = ' ';
            for (uint32_t p = start[w]; p < start[w + 1] && o < n; ++p) out[o++] = letters[p];
        }
        if (o < n) out[o++] = '\n';
    }
}

void gen_runs(uint8_t *out, uint64_t n, uint64_t chunk)
{
    Xs64 g{kSeed + chunk};
    uint64_t o = 0;
    while (o < n) {
        const uint8_t sym = (uint8_t)('a' + (g.nx() & 1u));
        const uint32_t len = 1 + g.nx() % 8192;
        for (uint32_t k = 0; k < len && o < n; ++k) out[o++] = sym;
        if (o < n) out[o++] = '\n';
    }
}

void gen_periodic(uint8_t *out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = (i % 4096 == 4095) ? '\n' : 'a';
}

}  // namespace

extern "C" int pss_gen_corpus(int kind, uint8_t *out, uint64_t n, uint64_t chunk_index)
{
    if (!out && n) return PSS_EINVAL;
    switch (kind) {
        case PSS_CORPUS_LINES: gen_lines(out, n, chunk_index); break;
        case PSS_CORPUS_WORDS: gen_words(out, n, chunk_index); break;
        case PSS_CORPUS_RUNS: gen_runs(out, n, chunk_index); break;
        case PSS_CORPUS_PERIODIC: gen_periodic(out, n); break;
        default: return PSS_EINVAL;
    }
    if (n) out[n - 1] = '\n';
    return PSS_OK;
}
