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

// Source-like text (round 5): what users of the library index most -- program sources, headers, documentation -- and the
// slowest regime of the builder (repeats of every length, 200-odd byte values, a fifth of the lines starting with blanks).
// The model, measured against 412 MB of real files (tests/tools/lcp_stats.py prints the same figures for both):
//   * "files" of 20 .. 3000 lines; most open with one of six licence headers (comment lines, one of them with a year and
//     a name that differ from file to file), one file in six is a copy of an earlier file with a few lines edited;
//   * lines: blank / comment / rule (one of "-=*~#" 40 .. 79 times) / import, include / def, class, function head /
//     statement / table of hexadecimal numbers / string of bytes above 127 / closing line; indentation follows a random
//     walk over 0 .. 7 levels of four blanks (one file in eight: tabs); identifiers from a 32768-word vocabulary with
//     the skew of `words`, keywords from a fixed list;
//   * inside a file, stretches of 2 .. 120 lines are copied from anywhere earlier in the chunk at line granularity
//     (copies of copies included: the repeat counts come out heavy-tailed), and stretches of 3 .. 40 lines from a pool of
//     512 boilerplate blocks that is the same for every chunk.
// Integer-only (xorshift64 as above), one thread.
struct SourceGen {
    uint8_t *out;
    uint64_t n, o = 0;
    Xs64 g;
    std::vector<uint8_t> letters;          // vocabulary
    std::vector<uint32_t> wstart;
    std::vector<uint64_t> line_start;      // start of every line written so far
    std::vector<uint64_t> file_start;      // index into line_start of every file's first line
    std::vector<std::vector<uint8_t>> pool, headers;
    int depth = 0;
    bool tabs = false;

    static constexpr uint32_t V = 32768;
    // What happens next inside a file, out of 1000: new lines / a line or two copied / a function's worth copied / a block
    // copied / boilerplate from the pool; one file in P_FILECOPY is a copy of an earlier file (a quarter of those
    // verbatim, the others with one line in FILE_EDIT written anew); indentation up to MAXD levels; P_RULE - 20 lines
    // in 100 are rules.  Fitted with tests/tools/lcp_stats.py to files found on the development machine (492 MB of
    // Python, C / C++ headers, documentation): tied after 12 / 20 / 37 / 64 / 512 / 4096 symbols there 89 / 74 / 56 / 44 /
    // 18 / 7 %, members of groups above 512 / 4096 suffixes at depth 20: 16 / 7 %; here, at 64 MiB, 82 / 69 / 53 / 45 /
    // 23 / 6 % and 15 / 7 %.
    static constexpr uint32_t E_FRESH = 480, E_LINE = 910, E_FUNC = 975, E_BLOCK = 979, P_FILECOPY = 12, FILE_EDIT = 120, MAXD = 5,
                              P_RULE = 21;

    void build_vocabulary()
    {
        static const char L1[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ_";
        static const char L2[] = "abcdefghijklmnopqrstuvwxyzabcdefghijklmnopqrstuvwxyz_0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
        Xs64 v{0x2545F4914F6CDD1DULL ^ 0x5DEECE66DULL};
        wstart.resize(V + 1);
        for (uint32_t w = 0; w < V; ++w) {
            wstart[w] = (uint32_t)letters.size();
            const uint32_t len = 2 + v.nx() % 11;
            const uint32_t cap = v.nx() % 8;      // one word in eight starts with a capital / underscore
            letters.push_back((uint8_t)L1[cap == 0 ? 26 + v.nx() % 27 : v.nx() % 26]);
            for (uint32_t k = 1; k < len; ++k) letters.push_back((uint8_t)L2[v.nx() % (sizeof(L2) - 1)]);
        }
        wstart[V] = (uint32_t)letters.size();
    }

    // ---- emitters into a byte vector (a line is built first, then written: the pool and the headers use the same code)
    void put(std::vector<uint8_t> &l, const char *s) { while (*s) l.push_back((uint8_t)*s++); }
    void word(std::vector<uint8_t> &l, Xs64 &r)
    {
        const uint32_t a = r.nx() % V, sh = r.nx() % 15;
        const uint32_t w = a >> sh;
        l.insert(l.end(), letters.begin() + wstart[w], letters.begin() + wstart[w + 1]);
    }
    void number(std::vector<uint8_t> &l, Xs64 &r)
    {
        uint32_t v = r.nx() % 100000u;
        if (r.nx() % 4 == 0) v %= 10;
        char buf[16];
        int k = 0;
        do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) l.push_back((uint8_t)buf[--k]);
    }
    void indent(std::vector<uint8_t> &l, int d)
    {
        for (int i = 0; i < d; ++i) {
            if (tabs) l.push_back('\t');
            else put(l, "    ");
        }
    }
    void expr(std::vector<uint8_t> &l, Xs64 &r, int budget)
    {
        switch (r.nx() % 8) {
            case 0: number(l, r); break;
            case 1: {
                l.push_back('\'');
                word(l, r);
                l.push_back('\'');
                break;
            }
            case 2:
            case 3: {
                word(l, r);
                l.push_back('(');
                const uint32_t args = budget > 0 ? r.nx() % 4 : 0;
                for (uint32_t a = 0; a < args; ++a) {
                    if (a) put(l, ", ");
                    expr(l, r, budget - 1);
                }
                l.push_back(')');
                break;
            }
            case 4: {
                word(l, r);
                l.push_back('.');
                word(l, r);
                break;
            }
            case 5: {
                word(l, r);
                static const char *ops[] = {" + ", " - ", " * ", " == ", " != ", " < ", " and ", " or ", " | ", " & ", " >> "};
                put(l, ops[r.nx() % 11]);
                if (budget > 0) expr(l, r, budget - 1);
                else number(l, r);
                break;
            }
            case 6: {
                word(l, r);
                l.push_back('[');
                if (budget > 0) expr(l, r, budget - 1);
                else number(l, r);
                l.push_back(']');
                break;
            }
            default: word(l, r); break;
        }
    }
    // one line (without the newline); may change the depth
    void make_line(std::vector<uint8_t> &l, Xs64 &r, int &d)
    {
        static const char *kw_open[] = {"if ", "for ", "while ", "elif ", "with ", "switch ", "else if "};
        static const char *types[] = {"int", "void", "const char *", "uint32_t", "size_t", "bool", "double", "auto", "static int", "unsigned"};
        const uint32_t k = r.nx() % 100;
        if (k < 8) return;                                       // blank
        if (k < 20) {                                            // comment
            indent(l, d);
            put(l, r.nx() % 2 ? "# " : "// ");
            const uint32_t words = 2 + r.nx() % 10;
            for (uint32_t i = 0; i < words; ++i) {
                if (i) l.push_back(' ');
                word(l, r);
            }
            if (r.nx() % 3 == 0) l.push_back('.');
            return;
        }
        if (k < P_RULE) {                                            // rule
            indent(l, d);
            static const char *lead[] = {"", "# ", "// ", "/* "};
            put(l, lead[r.nx() % 4]);
            const uint8_t c = (uint8_t)"-=*~#"[r.nx() % 5];
            const uint32_t len = 40 + r.nx() % 40;
            l.insert(l.end(), len, c);
            return;
        }
        if (k < 30) {                                            // import / include
            switch (r.nx() % 3) {
                case 0: put(l, "import "); word(l, r); if (r.nx() % 2) { l.push_back('.'); word(l, r); } break;
                case 1: put(l, "from "); word(l, r); l.push_back('.'); word(l, r); put(l, " import "); word(l, r); break;
                default: put(l, "#include <"); word(l, r); l.push_back('/'); word(l, r); put(l, ".h>"); break;
            }
            return;
        }
        if (k < 38) {                                            // def / class / function head: one level deeper
            indent(l, d);
            const uint32_t f = r.nx() % 3;
            if (f == 0) { put(l, "def "); word(l, r); put(l, "(self"); }
            else if (f == 1) { put(l, "class "); word(l, r); l.push_back('('); word(l, r); }
            else { put(l, types[r.nx() % 10]); l.push_back(' '); word(l, r); l.push_back('('); put(l, types[r.nx() % 10]); l.push_back(' '); word(l, r); }
            const uint32_t args = r.nx() % 4;
            for (uint32_t a = 0; a < args; ++a) {
                put(l, ", ");
                if (f == 2) { put(l, types[r.nx() % 10]); l.push_back(' '); }
                word(l, r);
                if (f == 0 && r.nx() % 3 == 0) put(l, "=None");
            }
            put(l, f == 2 ? ") {" : "):");
            if (d < (int)MAXD) ++d;
            return;
        }
        if (k < 46) {                                            // control statement: one level deeper
            indent(l, d);
            put(l, kw_open[r.nx() % 7]);
            expr(l, r, 2);
            put(l, tabs ? ") {" : ":");
            if (d < (int)MAXD) ++d;
            return;
        }
        if (k < 82) {                                            // statement
            indent(l, d);
            switch (r.nx() % 6) {
                case 0: put(l, "return "); expr(l, r, 2); break;
                case 1: put(l, "self."); word(l, r); put(l, " = "); expr(l, r, 2); break;
                case 2: expr(l, r, 2); break;
                default: word(l, r); put(l, " = "); expr(l, r, 2); break;
            }
            if (tabs) l.push_back(';');
            return;
        }
        if (k < 86) {                                            // table of numbers
            indent(l, d);
            static const char HEX[] = "0123456789abcdef";
            const uint32_t cnt = 4 + r.nx() % 9;
            for (uint32_t i = 0; i < cnt; ++i) {
                put(l, "0x");
                uint32_t v = r.nx();
                if (r.nx() % 3 == 0) v &= 0xffu;
                for (int s = 28; s >= 0; s -= 4) l.push_back((uint8_t)HEX[(v >> s) & 15u]);
                put(l, i + 1 < cnt ? ", " : ",");
            }
            return;
        }
        if (k < 89) {                                            // text with bytes above 127
            indent(l, d);
            word(l, r);
            put(l, " = \"");
            const uint32_t cnt = 4 + r.nx() % 28;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t x = r.nx();
                if (x % 4 == 0) l.push_back(' ');
                else l.push_back((uint8_t)(0x80u + (x >> 8) % 128u));
            }
            l.push_back('"');
            return;
        }
        // closing line: one level up
        if (d > 0) --d;
        indent(l, d);
        static const char *close[] = {"}", "pass", "return", "break", "};", "continue", "#endif", "return None"};
        put(l, tabs ? "}" : close[r.nx() % 8]);
    }

    bool full() const { return o >= n; }
    void write_line(const std::vector<uint8_t> &l)
    {
        if (full()) return;
        line_start.push_back(o);
        const uint64_t len = std::min<uint64_t>(l.size(), n - o);
        memcpy(out + o, l.data(), len);
        o += len;
        if (o < n) out[o++] = '\n';
    }
    void write_lines(const std::vector<uint8_t> &block)   // a block holds whole lines, each with its newline
    {
        size_t a = 0;
        while (a < block.size() && !full()) {
            size_t e = a;
            while (block[e] != '\n') ++e;
            line_start.push_back(o);
            const uint64_t len = std::min<uint64_t>(e + 1 - a, n - o);
            memcpy(out + o, block.data() + a, len);
            o += len;
            a = e + 1;
        }
    }
    // lines [first, first + count) of what has been written, copied to the end (the source may run into the copy)
    void copy_lines(uint64_t first, uint64_t count)
    {
        const uint64_t have = line_start.size();
        for (uint64_t i = 0; i < count && first + i + 1 < have && !full(); ++i) {
            const uint64_t a = line_start[first + i], e = line_start[first + i + 1];
            line_start.push_back(o);
            const uint64_t len = std::min<uint64_t>(e - a, n - o);
            memmove(out + o, out + a, len);
            o += len;
        }
    }

    void build_pools()
    {
        Xs64 r{0xA0761D6478BD642FULL};
        std::vector<uint8_t> l;
        headers.resize(6);
        for (auto &h : headers) {
            const uint32_t lines = 6 + r.nx() % 22;
            const bool slash = r.nx() % 2;
            for (uint32_t i = 0; i < lines; ++i) {
                put(h, slash ? "//" : "#");
                const uint32_t words = r.nx() % 13;
                for (uint32_t k = 0; k < words; ++k) { h.push_back(' '); word(h, r); }
                h.push_back('\n');
            }
        }
        pool.resize(512);
        for (auto &p : pool) {
            const uint32_t lines = 3 + r.nx() % 38;
            int d = (int)(r.nx() % 3);
            for (uint32_t i = 0; i < lines; ++i) {
                l.clear();
                make_line(l, r, d);
                p.insert(p.end(), l.begin(), l.end());
                p.push_back('\n');
            }
        }
    }

    void run()
    {
        build_vocabulary();
        build_pools();
        std::vector<uint8_t> l;
        while (!full()) {
            // ---- one file
            const uint64_t my_first_line = line_start.size();
            const uint32_t kind = g.nx() % P_FILECOPY;
            if (kind == 0 && file_start.size() >= 2) {
                // a copy of an earlier file, a line in sixty-four written anew
                const uint64_t f = g.nx() % (file_start.size() - 1);
                const uint64_t a = file_start[f], e = file_start[f + 1];
                int d = 0;
                const bool verbatim = g.nx() % 4 == 0;              // (vendored copies: not a byte changed)
                for (uint64_t i = a; i < e && !full(); ++i) {
                    if (!verbatim && g.nx() % FILE_EDIT == 0) {
                        l.clear();
                        make_line(l, g, d);
                        write_line(l);
                    } else copy_lines(i, 1);
                }
                file_start.push_back(my_first_line);
                continue;
            }
            tabs = g.nx() % 8 == 0;
            depth = 0;
            if (g.nx() % 4 != 0) {
                const auto &h = headers[g.nx() % 6];
                l.clear();
                put(l, h[0] == '/' ? "// Copyright (c) " : "# Copyright (c) ");
                { uint32_t y = 1990 + g.nx() % 36; char b4[4]; for (int i = 3; i >= 0; --i) { b4[i] = (char)('0' + y % 10); y /= 10; } l.insert(l.end(), b4, b4 + 4); }
                l.push_back(' ');
                word(l, g);
                l.push_back(' ');
                word(l, g);
                write_line(l);
                write_lines(h);
            }
            uint32_t body = 20 + g.nx() % 600;
            if (g.nx() % 8 == 0) body *= 5;
            for (uint32_t done = 0; done < body && !full();) {
                const uint32_t ev = g.nx() % 1000;
                const bool can_copy = line_start.size() > 64;
                uint32_t len = 0;
                if (ev >= E_FRESH && can_copy) {
                    if (ev < E_LINE) len = 1 + g.nx() % 2;                 // a line or two seen before
                    else if (ev < E_FUNC) len = 3 + g.nx() % 10;           // a function's worth
                    else if (ev < E_BLOCK) len = 12 + g.nx() % 89;         // a block
                }
                if (len) {
                    const uint64_t first = (((uint64_t)g.nx() << 32) | g.nx()) % (line_start.size() - 1);
                    copy_lines(first, len);
                    done += len;
                } else if (ev >= E_BLOCK) {                              // boilerplate, the same in every chunk
                    write_lines(pool[(g.nx() % 512) >> (g.nx() % 6)]);
                    done += 20;
                } else {                                                 // 1 .. 12 new lines
                    const uint32_t fresh = 1 + g.nx() % 12;
                    for (uint32_t i = 0; i < fresh && !full(); ++i) {
                        l.clear();
                        make_line(l, g, depth);
                        write_line(l);
                    }
                    done += fresh;
                }
            }
            file_start.push_back(my_first_line);
        }
    }
};

void gen_source(uint8_t *out, uint64_t n, uint64_t chunk)
{
    SourceGen sg;
    sg.out = out;
    sg.n = n;
    sg.g = Xs64{(kSeed ^ 0x8EBC6AF09C88C6E3ULL) + chunk};
    sg.run();
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
        case PSS_CORPUS_SOURCE: gen_source(out, n, chunk_index); break;
        default: return PSS_EINVAL;
    }
    if (n) out[n - 1] = '\n';
    return PSS_OK;
}
