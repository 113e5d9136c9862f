// corpus.cpp -- deterministic synthetic corpora for bench.py and the tests
// (generator spec: SURVEY.md section 8(d)).  Integer-only xorshift64, so any
// re-implementation (tests carry a Python one) agrees bit for bit.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
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

// xorshift64 is linear over GF(2): one step is a 64 x 64 bit matrix M (col[j] = image of bit j), so the
// state after k steps is M^k s.  `lines` draws exactly one step per byte, which lets the generator jump
// to the start of every block and fill the blocks on several threads -- same bytes as the serial loop.
struct BitMat {
    uint64_t col[64];
    uint64_t apply(uint64_t v) const
    {
        uint64_t r = 0;
        for (int j = 0; v; ++j, v >>= 1)
            if (v & 1u) r ^= col[j];
        return r;
    }
};

BitMat xs64_power(uint64_t k)
{
    BitMat result, base;
    for (int j = 0; j < 64; ++j) {
        result.col[j] = 1ull << j;
        Xs64 g{1ull << j};
        base.col[j] = g.step();
    }
    for (; k; k >>= 1) {
        if (k & 1u) {
            BitMat t;
            for (int j = 0; j < 64; ++j) t.col[j] = base.apply(result.col[j]);
            result = t;
        }
        BitMat sq;
        for (int j = 0; j < 64; ++j) sq.col[j] = base.apply(base.col[j]);
        base = sq;
    }
    return result;
}

void fill_lines(uint8_t *out, uint64_t count, uint64_t state)
{
    static const char ALPHA[] = "abcdefghijklmnopqrstuvwxyz0123456789 .";
    Xs64 g{state};
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t r = g.nx();
        out[i] = (r % 40 == 0) ? '\n' : (uint8_t)ALPHA[(r >> 8) % 38];
    }
}

void gen_lines(uint8_t *out, uint64_t n, uint64_t chunk)
{
    constexpr uint64_t kBlock = 1ull << 20;
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    unsigned nthreads = std::min<uint64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), blocks);
    if (blocks < 4 || nthreads < 2) {
        fill_lines(out, n, kSeed + chunk);
        return;
    }
    const BitMat jump = xs64_power(kBlock);
    std::vector<uint64_t> start(blocks);
    start[0] = kSeed + chunk;
    for (uint64_t b = 1; b < blocks; ++b) start[b] = jump.apply(start[b - 1]);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
            for (uint64_t b = t; b < blocks; b += nthreads)
                fill_lines(out + b * kBlock, std::min(kBlock, n - b * kBlock), start[b]);
        });
    for (auto &th : pool) th.join();
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

// One 40-byte line (39 symbols of the `lines` alphabet + newline) repeated: every suffix is tied with the ones a
// multiple of 40 bytes away for as long as the text lasts -- repeats with a period above 1 that are not runs of one byte.
void gen_repeat_line(uint8_t *out, uint64_t n, uint64_t chunk)
{
    uint8_t line[40];
    fill_lines(line, 40, kSeed + chunk);
    for (int i = 0; i < 39; ++i)
        if (line[i] == '\n') line[i] = ' ';
    line[39] = '\n';
    for (uint64_t i = 0; i < n; ++i) out[i] = line[i % 40];
}

// A 1 MiB block of `lines` text repeated, every copy after the first with 16 single-byte edits at pseudo-random
// places: long duplicated stretches (tens of KiB between the edits of two copies), nothing periodic at small scale.
void gen_dup_blocks(uint8_t *out, uint64_t n, uint64_t chunk)
{
    static const char ALPHA[] = "abcdefghijklmnopqrstuvwxyz0123456789 .";
    constexpr uint64_t kBlk = 1ull << 20;
    const uint64_t first = std::min(kBlk, n);
    fill_lines(out, first, kSeed + chunk);
    Xs64 g{(kSeed ^ 0xD1B54A32D192ED03ULL) + chunk};
    for (uint64_t o = kBlk; o < n; o += kBlk) {
        const uint64_t len = std::min(kBlk, n - o);
        memcpy(out + o, out, len);
        for (int e = 0; e < 16; ++e) {
            const uint64_t pos = g.nx() % kBlk;
            const uint8_t val = (uint8_t)ALPHA[g.nx() % 38];
            if (pos < len) out[o + pos] = val;
        }
    }
}

// Natural text with a repetitive middle: `words`, whose middle third is replaced -- first half by ONE 60-byte line written
// over and over (a periodic stretch inside a text that is not periodic), second half by 64 KiB blocks copied from the
// first third of the chunk (64 source blocks drawn at random: every block comes back about twenty times, LCPs of up to
// 64 KiB between suffixes that no run, period or whole-text detector sees).
void gen_mixed(uint8_t *out, uint64_t n, uint64_t chunk)
{
    gen_words(out, n, chunk);
    const uint64_t a = n / 3, b = 2 * (n / 3), mid = a + (b - a) / 2;
    if (b - a < 4096) return;
    uint8_t line[60];
    fill_lines(line, 60, (kSeed ^ 0x9E3779B97F4A7C15ULL) + chunk);
    for (int i = 0; i < 59; ++i)
        if (line[i] == '\n') line[i] = ' ';
    line[59] = '\n';
    for (uint64_t i = a; i < mid; ++i) out[i] = line[(i - a) % 60];
    constexpr uint64_t kBlk = 64ull << 10;
    const uint64_t sources = std::max<uint64_t>(1, std::min<uint64_t>(64, a / kBlk));
    const uint64_t blk = std::min(kBlk, a);                 // (tiny texts: whatever the first third holds)
    Xs64 g{(kSeed ^ 0xC2B2AE3D27D4EB4FULL) + chunk};
    for (uint64_t o = mid; o < b; o += blk) {
        const uint64_t src = (g.nx() % sources) * blk;
        memcpy(out + o, out + src, std::min(blk, b - o));
    }
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
        case PSS_CORPUS_REPEAT_LINE: gen_repeat_line(out, n, chunk_index); break;
        case PSS_CORPUS_DUP_BLOCKS: gen_dup_blocks(out, n, chunk_index); break;
        case PSS_CORPUS_MIXED: gen_mixed(out, n, chunk_index); break;
        default: return PSS_EINVAL;
    }
    if (n) out[n - 1] = '\n';
    return PSS_OK;
}
