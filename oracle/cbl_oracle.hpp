// cbl_oracle.hpp — CPU restatement of the reference's bulk k-mer insertion path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may build, load or call it, and there only as
// the checker / the timed CPU baseline. The shipped path (cbl_amd/, include/cblx.h) never links it.
//
// What it restates (all paths relative to /root/reference):
//   src/kmer.rs:11-24,61-72,94-96,110-112,133-135,293-348   nucleotide code, rolling pack, canonical, rev-comp
//   src/necklace/mod.rs:13-31                               necklace_pos / revert_necklace_pos (normative)
//   src/necklace/minimizer.rs:5-92                          LexMinQueue (monotone deque + tied-min positions)
//   src/necklace/queue.rs:14-118                            NecklaceQueue (fwd and REVERSE)
//   src/cbl.rs:16-32,65-67,181-206,239-289,328-339,433-449  constants, word packing, chunking, get_seq_words,
//                                                           insert_seq, |=
//   src/wordset/mod.rs:18-48,63-120,187-216,240-244,382-437 WordSet (insert, insert_batch, serde map)
//   src/wordset/set_ops.rs:123-157                          WordSet |=
//   src/trievec/mod.rs:9-136,170-220, set_ops.rs:43-71      TrieVec (Vec | Trie), insert_sorted_iter, |=
//   src/trie.rs:9-131                                       256-ary byte trie
//   src/sliced_int.rs:12-114                                truncated little-endian suffix ints
//   src/bitvector/mod.rs:16-85, cxx/rank_bv.h:14-42         dynamic rank bitvector (semantics only; see below)
//   src/bitvector/tiny/mod.rs:10-105                        256-bit node bitmap
//   cxx/tiered_vec.h:31-89                                  rank -> bucket-id dynamic array (semantics only)
//   src/cbl.rs:127-160, examples/cbl.rs:117-142             bincode 1.3 DefaultOptions+varint file format
//
// Third-party pieces absent from /root/reference, restated from their published behaviour:
//   * vigna/sux WordDynRankSel<FenwickByteL> (submodule, no SHA)  -> FenwickBV below (set returns the OLD bit,
//     rank(i) = ones in [0,i): pinned by src/bitvector/mod.rs:148-170).
//   * imartayan/tiered-vector Seq::Tiered (submodule, no SHA)     -> TieredVec32 below (2-level ring-buffer
//     blocks; semantics Vec<u32>::insert/get: pinned by src/ffi.rs:29-39).
//   * bincode "1.3" + serde "1.0" (Cargo.toml:9,13; no lockfile)   -> Writer/Reader below. The reference has
//     NO test that fixes a serialized byte, so the file format is PARITY-UNPINNED by reference tests; it is
//     restated from the Serialize impls cited above and cross-checked against an independent Python
//     restatement (oracle/pyref.py) and SURVEY.md Appendix B vectors.
//
// Pinning status: transform + set semantics are pinned against every KAT the reference's tests hold for
// this path (tests/test_oracle_kats.py). Serialized bytes: "parity unpinned" (see above).
//
// Deliberate deviations that do not change results: queue values are stored as u32 (9-bit minimizers)
// instead of T; suffixes are stored in the smallest of u32/u64/u128 instead of [u8; BYTES]; K and
// PREFIX_BITS are runtime values instead of const generics.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cbl_oracle {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------- constants (src/cbl.rs:16-32,65-67; build.rs:26-52)
struct Params {
    unsigned K = 0, PREFIX_BITS = 0;
    unsigned KMER_BITS = 0, POS_BITS = 0, WORD_BITS = 0, SUFFIX_BITS = 0, BYTES = 0, WIDTH = 0;
    static unsigned ilog2_npo2(unsigned v) {  // ilog2(next_power_of_two(v))
        unsigned l = 0;
        while ((1u << l) < v) ++l;
        return l;
    }
    Params() {}
    Params(unsigned k, unsigned pb) : K(k), PREFIX_BITS(pb) {
        KMER_BITS = 2 * K;
        POS_BITS = ilog2_npo2(KMER_BITS);
        WORD_BITS = KMER_BITS + POS_BITS;
        SUFFIX_BITS = WORD_BITS > PREFIX_BITS ? WORD_BITS - PREFIX_BITS : 0;  // saturating_sub
        BYTES = (SUFFIX_BITS + 7) / 8;
        WIDTH = KMER_BITS > 8 ? KMER_BITS - 8 : 0;  // queue_width: saturating_sub(M-1), M=9
    }
};
static const unsigned M_BITS = 9;          // src/cbl.rs:16
static const size_t CHUNK_SIZE = 2048;     // src/cbl.rs:67
static const size_t THRESHOLD = 1024;      // src/wordset/mod.rs:34

// ---------------------------------------------------------------- src/kmer.rs:11-24
inline int nuc_code(uint8_t b) {
    switch (b) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'T': case 't': return 2;
        case 'G': case 'g': return 3;
        default: return -1;
    }
}
inline uint8_t code_nuc(unsigned c) { return "ACTG"[c & 3]; }

template <class T> inline unsigned popcount_t(T x) {
    return (unsigned)__builtin_popcountll((uint64_t)x) +
           (sizeof(T) > 8 ? (unsigned)__builtin_popcountll((uint64_t)((u128)x >> 64)) : 0u);
}
template <class T> inline T mask_bits(unsigned bits) {
    return bits >= sizeof(T) * 8 ? ~(T)0 : (((T)1 << bits) - 1);
}

// src/kmer.rs:293-348: reverse the K 2-bit groups and complement (XOR 0b10) each.
template <class T> inline T rev_comp(T x, unsigned K) {
    T r = 0;
    for (unsigned i = 0; i < K; ++i) {
        r = (r << 2) | ((x & 3) ^ 2);
        x >>= 2;
    }
    return r;
}

// ---------------------------------------------------------------- src/necklace/mod.rs:13-31
template <class T> struct NeckPos {
    T necklace;
    unsigned pos;
    bool operator==(const NeckPos& o) const { return necklace == o.necklace && pos == o.pos; }
};

template <class T> inline NeckPos<T> necklace_pos(T word, unsigned BITS) {
    T necklace = word, rot = word;
    unsigned pos = 0;
    for (int i = (int)BITS - 1; i >= 0; --i) {
        rot = ((rot & 1) << (BITS - 1)) | (rot >> 1);
        if (rot <= necklace) {
            necklace = rot;
            pos = (unsigned)i;
        }
    }
    return {necklace, pos};
}
template <class T> inline T revert_necklace_pos(T necklace, unsigned pos, unsigned BITS) {
    return ((necklace << (BITS - pos)) & mask_bits<T>(BITS)) | (necklace >> pos);
}

// ---------------------------------------------------------------- small ring deque (std::collections::VecDeque stand-in)
template <class V, unsigned CAP = 128> struct RingDeque {
    V a[CAP];
    unsigned head = 0, n = 0;
    void clear() { head = 0; n = 0; }
    unsigned size() const { return n; }
    bool empty() const { return n == 0; }
    V& operator[](unsigned i) { return a[(head + i) & (CAP - 1)]; }
    const V& operator[](unsigned i) const { return a[(head + i) & (CAP - 1)]; }
    void push_back(const V& v) { a[(head + n) & (CAP - 1)] = v; ++n; }
    void push_front(const V& v) { head = (head + CAP - 1) & (CAP - 1); a[head] = v; ++n; }
    void pop_front() { head = (head + 1) & (CAP - 1); --n; }
    void truncate(unsigned i) { if (i < n) n = i; }
};

// ---------------------------------------------------------------- src/necklace/minimizer.rs:5-92
struct LexMinQueue {
    struct E { uint32_t val; uint32_t pos; };
    RingDeque<E> deq;
    RingDeque<uint32_t> min_pos;
    unsigned pos = 0, WIDTH = 1;
    explicit LexMinQueue(unsigned width = 1) : WIDTH(width) {}

    template <class F> void for_each_min_pos(F f) const {  // iter_min_pos :20-24
        for (unsigned i = 0; i < min_pos.size(); ++i) f((min_pos[i] + WIDTH - pos) % WIDTH);
    }
    void refill_min_pos() {
        while (min_pos.size() < deq.size() && deq[min_pos.size()].val == deq[0].val)
            min_pos.push_back(deq[min_pos.size()].pos);
    }
    template <class F> void insert_full(F vals) {  // :26-44, vals(p) for p in 0..WIDTH, consumed in reverse
        deq.clear();
        min_pos.clear();
        uint32_t minimizer = vals(WIDTH - 1);
        unsigned p = (pos + WIDTH - 1) % WIDTH;
        deq.push_front({minimizer, p});
        for (int q = (int)WIDTH - 2; q >= 0; --q) {
            uint32_t u = vals((unsigned)q);
            p = (p + WIDTH - 1) % WIDTH;
            if (u <= minimizer) {
                minimizer = u;
                deq.push_front({minimizer, p});
            }
        }
        refill_min_pos();
    }
    void insert(uint32_t u) {  // :46-63
        if (!deq.empty() && deq[0].pos == pos) {
            deq.pop_front();
            if (!min_pos.empty()) min_pos.pop_front();
        }
        unsigned i = deq.size();
        while (i > 0 && deq[i - 1].val > u) --i;
        deq.truncate(i);
        min_pos.truncate(i);
        deq.push_back({u, pos});
        refill_min_pos();
        pos = (pos + 1) % WIDTH;
    }
    void insert2(uint32_t u, uint32_t v) {  // :65-91
        unsigned next_pos = (pos + 1) % WIDTH;
        if (!deq.empty() && deq[0].pos == pos) {
            deq.pop_front();
            if (!min_pos.empty()) min_pos.pop_front();
        }
        if (!deq.empty() && deq[0].pos == next_pos) {
            deq.pop_front();
            if (!min_pos.empty()) min_pos.pop_front();
        }
        uint32_t w = std::min(u, v);
        unsigned i = deq.size();
        while (i > 0 && deq[i - 1].val > w) --i;
        deq.truncate(i);
        min_pos.truncate(i);
        if (u <= v) deq.push_back({u, pos});
        deq.push_back({v, next_pos});
        refill_min_pos();
        pos = (next_pos + 1) % WIDTH;
    }
};

// ---------------------------------------------------------------- src/necklace/queue.rs:14-118
template <class T, bool REVERSE> struct NecklaceQueue {
    unsigned BITS, WIDTH, M;
    T MASK;
    uint32_t MIN_MASK;
    T word = 0;
    LexMinQueue min_queue;

    NecklaceQueue(unsigned bits, unsigned width)
        : BITS(bits), WIDTH(width), M(bits - width + 1), MASK(mask_bits<T>(bits)),
          MIN_MASK((1u << (bits - width + 1)) - 1), min_queue(width) {
        if (width < 2 || width > 120) throw std::runtime_error("NecklaceQueue: unsupported WIDTH");
    }
    T rotation(unsigned p) const {  // :48-50
        if (p == 0) return word;    // (word >> BITS) is 0 for T wider than BITS
        return ((word << p) & MASK) | (word >> (BITS - p));
    }
    NeckPos<T> get_necklace_pos() const {  // :53-79
        bool have = false;
        T best = 0;
        unsigned bestp = 0;
        auto consider = [&](unsigned p) {
            T r = rotation(p);
            if (!have || r < best || (r == best && p < bestp)) { best = r; bestp = p; have = true; }
        };
        min_queue.for_each_min_pos([&](unsigned p) { consider(REVERSE ? WIDTH - 1 - p : p); });
        for (unsigned p = WIDTH; p < BITS; ++p) consider(p);
        return {best, bestp};
    }
    void insert_full(T w) {  // :82-96
        word = w & MASK;
        if (REVERSE)
            min_queue.insert_full([&](unsigned p) { return (uint32_t)(w >> p) & MIN_MASK; });
        else
            min_queue.insert_full([&](unsigned p) { return (uint32_t)(w >> (BITS - p - M)) & MIN_MASK; });
    }
    void insert(unsigned x) {  // :99-107
        if (REVERSE) {
            word = (word >> 1) | ((T)(x & 1) << (BITS - 1));
            min_queue.insert((uint32_t)(word >> (WIDTH - 1)));
        } else {
            word = ((word << 1) & MASK) | (T)(x & 1);
            min_queue.insert((uint32_t)word & MIN_MASK);
        }
    }
    void insert2(unsigned x) {  // :110-118
        if (REVERSE) {
            word = (word >> 2) | ((T)(x & 3) << (BITS - 2));
            min_queue.insert2((uint32_t)(word >> (WIDTH - 2)) & MIN_MASK, (uint32_t)(word >> (WIDTH - 1)));
        } else {
            word = ((word << 2) & MASK) | (T)(x & 3);
            min_queue.insert2((uint32_t)(word >> 1) & MIN_MASK, (uint32_t)word & MIN_MASK);
        }
    }
};

// ---------------------------------------------------------------- cxx/rank_bv.h:14-42 (semantics; Fenwick over word popcounts)
struct FenwickBV {
    std::vector<uint64_t> words;
    std::vector<uint32_t> fen;  // 1-based Fenwick tree over popcount(words[i])
    size_t nbits = 0;
    explicit FenwickBV(size_t size = 0) : words((size + 63) / 64, 0), fen((size + 63) / 64 + 1, 0), nbits(size) {}
    size_t size() const { return nbits; }
    size_t num_blocks() const { return words.size(); }
    bool get(size_t i) const { return (words[i >> 6] >> (i & 63)) & 1; }
    void fen_add(size_t w, int32_t d) {
        for (size_t j = w + 1; j < fen.size(); j += j & (~j + 1)) fen[j] = (uint32_t)((int32_t)fen[j] + d);
    }
    bool set(size_t i) {  // returns the OLD bit (rank_bv.h:30)
        bool old = get(i);
        if (!old) { words[i >> 6] |= 1ull << (i & 63); fen_add(i >> 6, 1); }
        return old;
    }
    bool clear(size_t i) {
        bool old = get(i);
        if (old) { words[i >> 6] &= ~(1ull << (i & 63)); fen_add(i >> 6, -1); }
        return old;
    }
    uint64_t rank(size_t i) const {  // ones in [0, i)
        uint64_t r = 0;
        for (size_t j = i >> 6; j > 0; j -= j & (~j + 1)) r += fen[j];
        if (i & 63) r += (uint64_t)__builtin_popcountll(words[i >> 6] & ((1ull << (i & 63)) - 1));
        return r;
    }
    uint64_t get_block(size_t b) const { return words[b]; }
    void update_block(size_t b, uint64_t v) {
        int32_t d = __builtin_popcountll(v) - __builtin_popcountll(words[b]);
        words[b] = v;
        if (d) fen_add(b, d);
    }
    size_t count_all() const { return (size_t)rank(nbits); }
    template <class F> void for_each_set(F f) const {  // src/bitvector/mod.rs:55-85 (ascending)
        for (size_t b = 0; b < words.size(); ++b) {
            uint64_t w = words[b];
            while (w) {
                unsigned t = (unsigned)__builtin_ctzll(w);
                w &= w - 1;
                f(b * 64 + t);
            }
        }
    }
};

// ---------------------------------------------------------------- cxx/tiered_vec.h:31-89 (semantics; 2-level ring blocks)
struct TieredVec32 {
    static const size_t B = 1024;
    struct Blk {
        uint32_t a[B];
        uint32_t head = 0, n = 0;
        uint32_t& at(size_t i) { return a[(head + i) & (B - 1)]; }
    };
    std::vector<std::unique_ptr<Blk>> blks;
    size_t length = 0;
    size_t len() const { return length; }
    uint32_t get(size_t i) const { return blks[i / B]->at(i % B); }
    void insert(size_t i, uint32_t v) {
        if (length == blks.size() * B) blks.emplace_back(new Blk());
        size_t b = i / B, o = i % B;
        Blk* k = blks[b].get();
        bool has = false;
        uint32_t carry = 0;
        if (k->n == B) { carry = k->at(B - 1); k->n--; has = true; }
        for (size_t j = k->n; j > o; --j) k->at(j) = k->at(j - 1);
        k->at(o) = v;
        k->n++;
        for (size_t j = b + 1; has; ++j) {
            Blk* q = blks[j].get();
            bool full = q->n == B;
            uint32_t c2 = 0;
            if (full) { c2 = q->at(B - 1); q->n--; }
            q->head = (q->head + B - 1) & (B - 1);
            q->a[q->head] = carry;
            q->n++;
            has = full;
            carry = c2;
        }
        ++length;
    }
};

// ---------------------------------------------------------------- src/bitvector/tiny/mod.rs + src/trie.rs
struct TrieNode {
    uint64_t bv[4] = {0, 0, 0, 0};
    std::vector<TrieNode*> children;
    ~TrieNode() { for (auto* c : children) delete c; }
    bool bv_contains(uint8_t i) const { return (bv[i >> 6] >> (i & 63)) & 1; }
    bool bv_insert(uint8_t i) {  // tiny/mod.rs:37-41 -> true if it was absent
        uint64_t old = bv[i >> 6];
        bv[i >> 6] = old | (1ull << (i & 63));
        return bv[i >> 6] != old;
    }
    unsigned bv_rank(uint8_t i) const {  // tiny/mod.rs:51-63
        unsigned r = (unsigned)__builtin_popcountll(bv[i >> 6] & ((1ull << (i & 63)) - 1));
        for (unsigned w = 0; w < (unsigned)(i >> 6); ++w) r += (unsigned)__builtin_popcountll(bv[w]);
        return r;
    }
    unsigned bv_count() const {
        return (unsigned)(__builtin_popcountll(bv[0]) + __builtin_popcountll(bv[1]) + __builtin_popcountll(bv[2]) +
                          __builtin_popcountll(bv[3]));
    }
    TrieNode* clone() const {
        TrieNode* n = new TrieNode();
        memcpy(n->bv, bv, sizeof(bv));
        n->children.reserve(children.size());
        for (auto* c : children) n->children.push_back(c->clone());
        return n;
    }
};

// src/trievec/mod.rs:9-15 — S is the suffix storage int (u32/u64/u128), BYTES the serialized width.
template <class S> struct TrieVec {
    bool is_trie = false;
    std::vector<S> vec;
    TrieNode* trie = nullptr;
    size_t trie_len = 0;
    TrieVec() {}
    TrieVec(const TrieVec& o) : is_trie(o.is_trie), vec(o.vec), trie(o.trie ? o.trie->clone() : nullptr), trie_len(o.trie_len) {}
    TrieVec(TrieVec&& o) noexcept : is_trie(o.is_trie), vec(std::move(o.vec)), trie(o.trie), trie_len(o.trie_len) { o.trie = nullptr; }
    TrieVec& operator=(TrieVec&& o) noexcept {
        if (this != &o) { delete trie; is_trie = o.is_trie; vec = std::move(o.vec); trie = o.trie; trie_len = o.trie_len; o.trie = nullptr; }
        return *this;
    }
    TrieVec& operator=(const TrieVec& o) { if (this != &o) { TrieVec t(o); *this = std::move(t); } return *this; }
    ~TrieVec() { delete trie; }
    size_t len() const { return is_trie ? trie_len : vec.size(); }

    static inline uint8_t be_byte(S x, unsigned BYTES, unsigned d) { return (uint8_t)(x >> (8 * (BYTES - 1 - d))); }  // sliced_int.rs:50-54

    static bool trie_insert(TrieNode* t, S x, unsigned BYTES) {  // src/trie.rs:118-131
        for (unsigned d = 0; d + 1 < BYTES; ++d) {
            uint8_t idx = be_byte(x, BYTES, d);
            bool absent = t->bv_insert(idx);
            unsigned r = t->bv_rank(idx);
            if (absent) t->children.insert(t->children.begin() + r, new TrieNode());
            t = t->children[r];
        }
        return t->bv_insert(be_byte(x, BYTES, BYTES - 1));
    }
    static bool trie_contains(const TrieNode* t, S x, unsigned BYTES) {  // src/trie.rs:104-116
        for (unsigned d = 0; d + 1 < BYTES; ++d) {
            uint8_t idx = be_byte(x, BYTES, d);
            if (!t->bv_contains(idx)) return false;
            t = t->children[t->bv_rank(idx)];
        }
        return t->bv_contains(be_byte(x, BYTES, BYTES - 1));
    }
    template <class F> static void trie_iter(const TrieNode* t, unsigned BYTES, unsigned d, S acc, F& f) {  // numeric order
        for (unsigned w = 0, r = 0; w < 4; ++w) {
            uint64_t m = t->bv[w];
            while (m) {
                unsigned b = (unsigned)__builtin_ctzll(m);
                m &= m - 1;
                S v = acc | ((S)(w * 64 + b) << (8 * (BYTES - 1 - d)));
                if (d + 1 == BYTES) f(v);
                else trie_iter(t->children[r], BYTES, d + 1, v, f);
                ++r;
            }
        }
    }

    bool contains(S x, unsigned BYTES) const {  // :65-70
        if (is_trie) return trie_contains(trie, x, BYTES);
        return std::find(vec.begin(), vec.end(), x) != vec.end();
    }
    bool insert(S x, unsigned BYTES) {  // :72-89
        if (is_trie) {
            bool absent = trie_insert(trie, x, BYTES);
            if (absent) ++trie_len;
            return absent;
        }
        if (std::find(vec.begin(), vec.end(), x) == vec.end()) { vec.push_back(x); return true; }
        return false;
    }
    void as_trie(unsigned BYTES) {  // :170-178
        if (is_trie) return;
        trie = new TrieNode();
        for (S x : vec) trie_insert(trie, x, BYTES);
        trie_len = vec.size();
        is_trie = true;
        std::vector<S>().swap(vec);
    }
    // iter_sorted (:209-220): sorts a Vec IN PLACE, then yields ascending.
    std::vector<S> iter_sorted(unsigned BYTES) {
        if (!is_trie) { std::sort(vec.begin(), vec.end()); return vec; }
        std::vector<S> out;
        out.reserve(trie_len);
        auto f = [&](S v) { out.push_back(v); };
        trie_iter(trie, BYTES, 0, (S)0, f);
        return out;
    }
    std::vector<S> iter_stored(unsigned BYTES) const {  // iter (:198-207): Vec in stored order, Trie ascending
        if (!is_trie) return vec;
        std::vector<S> out;
        out.reserve(trie_len);
        auto f = [&](S v) { out.push_back(v); };
        trie_iter(trie, BYTES, 0, (S)0, f);
        return out;
    }
    void insert_sorted_iter(const std::vector<S>& it, unsigned BYTES) {  // :118-136
        if (is_trie) { for (S x : it) insert(x, BYTES); return; }
        size_t stop = vec.size(), i = 0;
        for (S x : it) {
            while (i < stop && x > vec[i]) ++i;
            if (i == stop || x < vec[i]) vec.push_back(x);
        }
    }
    void bitor_assign(TrieVec& other, unsigned BYTES) {  // src/trievec/set_ops.rs:43-71
        std::vector<S> a = iter_sorted(BYTES), b = other.iter_sorted(BYTES), ins;
        size_t i = 0, j = 0;
        while (i < a.size() && j < b.size()) {
            if (a[i] < b[j]) ++i;
            else if (a[i] > b[j]) ins.push_back(b[j++]);
            else { ++i; ++j; }
        }
        while (j < b.size()) ins.push_back(b[j++]);
        insert_sorted_iter(ins, BYTES);
    }
};

// ---------------------------------------------------------------- bincode 1.3 DefaultOptions (varint, LE)
struct Writer {
    std::vector<uint8_t> out;
    void u8(uint8_t b) { out.push_back(b); }
    void varint(uint64_t v) {
        if (v <= 250) { out.push_back((uint8_t)v); }
        else if (v < (1ull << 16)) { out.push_back(0xFB); for (int i = 0; i < 2; ++i) out.push_back((uint8_t)(v >> (8 * i))); }
        else if (v < (1ull << 32)) { out.push_back(0xFC); for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(v >> (8 * i))); }
        else { out.push_back(0xFD); for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(v >> (8 * i))); }
    }
};
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    Reader(const uint8_t* d, size_t n) : p(d), end(d + n) {}
    uint8_t u8() { if (p >= end) throw std::runtime_error("bincode: unexpected EOF"); return *p++; }
    uint64_t varint() {
        uint8_t t = u8();
        if (t <= 250) return t;
        unsigned nb = t == 0xFB ? 2 : t == 0xFC ? 4 : t == 0xFD ? 8 : 0;
        if (!nb) throw std::runtime_error("bincode: bad varint tag");
        uint64_t v = 0;
        for (unsigned i = 0; i < nb; ++i) v |= (uint64_t)u8() << (8 * i);
        return v;
    }
};

// ---------------------------------------------------------------- src/wordset/mod.rs
template <class T, class S> struct WordSet {
    Params P;
    FenwickBV prefixes;
    TieredVec32 tiered;
    std::vector<TrieVec<S>> suffix_containers;
    std::vector<size_t> empty_containers;
    std::vector<std::pair<size_t, S>> scratch;  // the per-chunk Vec<(usize, SlicedInt)> (:188-191)

    explicit WordSet(const Params& p) : P(p), prefixes((size_t)1 << p.PREFIX_BITS) {
        if (p.PREFIX_BITS > 32) throw std::runtime_error("PREFIX_BITS should be <= 32");
        if (p.SUFFIX_BITS == 0) throw std::runtime_error("SUFFIX_BITS should be != 0");
    }
    size_t count() const { size_t c = 0; for (auto& b : suffix_containers) c += b.len(); return c; }
    bool is_empty() const { return prefixes.count_all() == 0; }
    inline void split(T word, size_t& prefix, S& suffix) const {  // :63-71
        prefix = (size_t)(word >> P.SUFFIX_BITS);
        suffix = (S)(word & mask_bits<T>(P.SUFFIX_BITS));
    }
    bool contains(T word) const {  // :86-95
        size_t prefix; S suffix;
        split(word, prefix, suffix);
        if (!prefixes.get(prefix)) return false;
        size_t id = tiered.get((size_t)prefixes.rank(prefix));
        return suffix_containers[id].contains(suffix, P.BYTES);
    }
    void adapt_container_grow(size_t id) {  // :240-244
        if (suffix_containers[id].len() > THRESHOLD) suffix_containers[id].as_trie(P.BYTES);
    }
    bool insert(T word) {  // :97-120
        size_t prefix; S suffix;
        split(word, prefix, suffix);
        bool absent = !prefixes.set(prefix);
        size_t rank = (size_t)prefixes.rank(prefix);
        if (absent) {
            size_t id = suffix_containers.size();
            suffix_containers.emplace_back();
            suffix_containers[id].vec.push_back(suffix);
            tiered.insert(rank, (uint32_t)id);
        } else {
            size_t id = tiered.get(rank);
            absent = suffix_containers[id].insert(suffix, P.BYTES);
            adapt_container_grow(id);
        }
        return absent;
    }
    void insert_batch(const T* words, size_t n) {  // :187-216
        scratch.clear();
        scratch.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            size_t p; S s;
            split(words[i], p, s);
            scratch.emplace_back(p, s);
        }
        size_t i = 0;
        while (i < n) {  // chunk_by equal consecutive prefixes
            size_t j = i + 1;
            while (j < n && scratch[j].first == scratch[i].first) ++j;
            size_t prefix = scratch[i].first;
            bool absent = !prefixes.set(prefix);
            size_t rank = (size_t)prefixes.rank(prefix);
            size_t id;
            if (absent) {
                id = suffix_containers.size();
                suffix_containers.emplace_back();
                tiered.insert(rank, (uint32_t)id);
            } else {
                id = tiered.get(rank);
            }
            for (size_t t = i; t < j; ++t) suffix_containers[id].insert(scratch[t].second, P.BYTES);
            adapt_container_grow(id);
            i = j;
        }
    }
    // src/wordset/set_ops.rs:123-157
    void bitor_assign(WordSet& other) {
        std::vector<size_t> mine;
        prefixes.for_each_set([&](size_t p) { mine.push_back(p); });
        size_t it = 0, rank = 0, other_rank = 0;
        other.prefixes.for_each_set([&](size_t other_prefix) {
            while (it < mine.size() && mine[it] < other_prefix) { ++it; ++rank; }
            size_t other_id = other.tiered.get(other_rank);
            if (it < mine.size() && mine[it] == other_prefix) {
                size_t id = tiered.get(rank);
                suffix_containers[id].bitor_assign(other.suffix_containers[other_id], P.BYTES);
                ++it; ++rank;
            } else {
                size_t id = suffix_containers.size();
                suffix_containers.push_back(other.suffix_containers[other_id]);
                tiered.insert(rank, (uint32_t)id);
                ++rank;
            }
            ++other_rank;
        });
        for (size_t b = 0; b < prefixes.num_blocks(); ++b)  // src/bitvector/set_ops.rs:19-28
            prefixes.update_block(b, prefixes.get_block(b) | other.prefixes.get_block(b));
    }

    // ---- Serialize (:382-396) over derived TrieVec/Trie/TrieNode + SlicedInt (:110-114) + TinyBitvector (:97-105)
    void ser_node(Writer& w, const TrieNode* t) const {
        w.varint(t->bv_count());
        for (unsigned wd = 0; wd < 4; ++wd) {
            uint64_t m = t->bv[wd];
            while (m) { unsigned b = (unsigned)__builtin_ctzll(m); m &= m - 1; w.u8((uint8_t)(wd * 64 + b)); }
        }
        w.varint(t->children.size());
        for (auto* c : t->children) ser_node(w, c);
    }
    void ser_bucket(Writer& w, const TrieVec<S>& b) const {
        if (!b.is_trie) {
            w.varint(0);
            w.varint(b.vec.size());
            for (S x : b.vec) {
                w.varint(P.BYTES);
                for (unsigned i = 0; i < P.BYTES; ++i) w.u8((uint8_t)(x >> (8 * i)));
            }
        } else {
            w.varint(1);
            ser_node(w, b.trie);
            w.varint(b.trie_len);
        }
    }
    void serialize(Writer& w) const {
        w.varint(tiered.len());
        size_t rank = 0;
        prefixes.for_each_set([&](size_t prefix) {
            w.varint((uint32_t)prefix);
            ser_bucket(w, suffix_containers[tiered.get(rank)]);
            ++rank;
        });
    }
    // ---- Deserialize (:398-437)
    TrieNode* de_node(Reader& r) const {
        std::unique_ptr<TrieNode> t(new TrieNode());
        uint64_t c = r.varint();
        for (uint64_t i = 0; i < c; ++i) t->bv_insert(r.u8());
        uint64_t nc = r.varint();
        t->children.reserve(nc);
        for (uint64_t i = 0; i < nc; ++i) t->children.push_back(de_node(r));
        return t.release();
    }
    void deserialize(Reader& r) {
        uint64_t n = r.varint();
        for (uint64_t e = 0; e < n; ++e) {
            size_t prefix = (size_t)(uint32_t)r.varint();
            TrieVec<S> b;
            uint64_t tag = r.varint();
            if (tag == 0) {
                uint64_t cnt = r.varint();
                b.vec.reserve(cnt);
                for (uint64_t i = 0; i < cnt; ++i) {
                    uint64_t nb = r.varint();
                    S x = 0;
                    for (uint64_t k = 0; k < nb; ++k) { uint8_t by = r.u8(); if (k < P.BYTES) x |= (S)by << (8 * k); }
                    b.vec.push_back(x);
                }
            } else if (tag == 1) {
                b.is_trie = true;
                b.trie = de_node(r);
                b.trie_len = (size_t)r.varint();
            } else throw std::runtime_error("bincode: bad TrieOrVec tag");
            size_t rank = suffix_containers.size();
            prefixes.set(prefix);
            tiered.insert(rank, (uint32_t)rank);
            suffix_containers.push_back(std::move(b));
        }
    }
};

// ---------------------------------------------------------------- src/cbl.rs
template <class T, class S> struct CBL {
    Params P;
    bool canonical;
    WordSet<T, S> wordset;
    NecklaceQueue<T, false> queue;
    NecklaceQueue<T, true> queue_rev;
    std::vector<T> words, words_rc;

    CBL(unsigned K, unsigned PB, bool canon)
        : P(K, PB), canonical(canon), wordset(P), queue(P.KMER_BITS, P.WIDTH), queue_rev(P.KMER_BITS, P.WIDTH) {
        if (P.WORD_BITS > sizeof(T) * 8) throw std::runtime_error("Cannot fit a K-mer and its length in T");  // :87-91
    }
    T merge_necklace_pos(T necklace, unsigned pos) const { return (necklace << P.POS_BITS) | (T)pos; }  // :181-184
    T get_word_bruteforce(T kmer) const {  // :199-206
        if (canonical && (popcount_t(kmer) & 1)) kmer = rev_comp(kmer, P.K);
        NeckPos<T> np = necklace_pos<T>(kmer, P.KMER_BITS);
        return merge_necklace_pos(np.necklace, np.pos);
    }
    T recover_kmer(T word) const {  // :208-214
        return revert_necklace_pos<T>(word >> P.POS_BITS, (unsigned)(word & mask_bits<T>(P.POS_BITS)), P.KMER_BITS);
    }
    // get_seq_words :247-289 (streaming queue path). Result left in `words`.
    void get_seq_words(const uint8_t* seq, size_t len) {
        const unsigned K = P.K;
        const T MASK = mask_bits<T>(P.KMER_BITS);
        words.clear();
        T kmer = 0;  // from_nucs(&seq[..K]): valid bases among the first K BYTES, no masking (kmer.rs:110-112,133-135)
        for (size_t i = 0; i < K; ++i) { int c = nuc_code(seq[i]); if (c >= 0) kmer = (kmer << 2) | (T)c; }
        if (canonical) {
            words_rc.clear();
            queue.insert_full(kmer);
            queue_rev.insert_full(rev_comp(kmer, K));
            auto emit = [&]() {
                if ((popcount_t(kmer) & 1) == 0) { auto np = queue.get_necklace_pos(); words.push_back(merge_necklace_pos(np.necklace, np.pos)); }
                else { auto np = queue_rev.get_necklace_pos(); words_rc.push_back(merge_necklace_pos(np.necklace, np.pos)); }
            };
            emit();
            for (size_t i = K; i < len; ++i) {
                int c = nuc_code(seq[i]);
                if (c < 0) continue;
                kmer = ((kmer << 2) | (T)c) & MASK;
                queue.insert2((unsigned)c);
                queue_rev.insert2((unsigned)c ^ 2u);
                emit();
            }
            words.insert(words.end(), words_rc.begin(), words_rc.end());
        } else {
            queue.insert_full(kmer);
            { auto np = queue.get_necklace_pos(); words.push_back(merge_necklace_pos(np.necklace, np.pos)); }
            for (size_t i = K; i < len; ++i) {
                int c = nuc_code(seq[i]);
                if (c < 0) continue;
                queue.insert2((unsigned)c);
                auto np = queue.get_necklace_pos();
                words.push_back(merge_necklace_pos(np.necklace, np.pos));
            }
        }
    }
    // Same output through the normative brute-force definition (used to pin queue == brute force).
    void get_seq_words_bruteforce(const uint8_t* seq, size_t len, std::vector<T>& out) const {
        const unsigned K = P.K;
        const T MASK = mask_bits<T>(P.KMER_BITS);
        std::vector<T> rc;
        out.clear();
        T kmer = 0;
        for (size_t i = 0; i < K; ++i) { int c = nuc_code(seq[i]); if (c >= 0) kmer = (kmer << 2) | (T)c; }
        auto emit = [&]() {
            if (!canonical) { auto np = necklace_pos<T>(kmer & MASK, P.KMER_BITS); out.push_back(merge_necklace_pos(np.necklace, np.pos)); }
            else if ((popcount_t(kmer) & 1) == 0) { auto np = necklace_pos<T>(kmer & MASK, P.KMER_BITS); out.push_back(merge_necklace_pos(np.necklace, np.pos)); }
            else { auto np = necklace_pos<T>(rev_comp(kmer, K), P.KMER_BITS); rc.push_back(merge_necklace_pos(np.necklace, np.pos)); }
        };
        emit();
        for (size_t i = K; i < len; ++i) {
            int c = nuc_code(seq[i]);
            if (c < 0) continue;
            kmer = ((kmer << 2) | (T)c) & MASK;
            emit();
        }
        out.insert(out.end(), rc.begin(), rc.end());
    }
    template <class F> void for_each_chunk(const uint8_t* seq, size_t len, F f) const {  // :239-243
        for (size_t start = 0; start < len - P.K + 1; start += CHUNK_SIZE)
            f(seq + start, std::min(start + CHUNK_SIZE + P.K - 1, len) - start);
    }
    void insert_seq(const uint8_t* seq, size_t len) {  // :328-339
        if (len < P.K) throw std::runtime_error("Sequence size (" + std::to_string(len) + ") is smaller than K (" + std::to_string(P.K) + ")");
        for_each_chunk(seq, len, [&](const uint8_t* c, size_t n) {
            get_seq_words(c, n);
            wordset.insert_batch(words.data(), words.size());
        });
    }
    bool insert_kmer(T kmer) { return wordset.insert(get_word_bruteforce(kmer)); }    // :226-228
    bool contains_kmer(T kmer) const { return wordset.contains(get_word_bruteforce(kmer)); }  // :219-221
    size_t count() const { return wordset.count(); }
    void bitor_assign(CBL& other) {  // :433-449
        if (canonical != other.canonical) throw std::runtime_error("One of the index is canonical while the other isn't");
        wordset.bitor_assign(other.wordset);
    }
    void serialize(Writer& w) const { w.u8(canonical ? 1 : 0); wordset.serialize(w); }  // derive :40-54
    void deserialize(const uint8_t* d, size_t n) {
        Reader r(d, n);
        canonical = r.u8() != 0;
        wordset.deserialize(r);
        if (r.p != r.end) throw std::runtime_error("bincode: trailing bytes");  // reject_trailing_bytes
    }
};

}  // namespace cbl_oracle
