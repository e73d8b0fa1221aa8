// cbl_oracle_capi.cpp — extern "C" surface of the CPU oracle, for ctypes (tests/, smoke(), bench cpu_baseline).
// TEST INFRASTRUCTURE ONLY — see the header of cbl_oracle.hpp. Never linked by the product library.
#include "cbl_oracle.hpp"

#include <chrono>

using namespace cbl_oracle;

namespace {

thread_local std::string g_err;

struct ICBL {
    virtual ~ICBL() {}
    virtual void insert_seq(const uint8_t* s, size_t n) = 0;
    virtual size_t count() const = 0;
    virtual void serialize(Writer& w) const = 0;
    virtual void load(const uint8_t* d, size_t n) = 0;
    virtual void merge(ICBL* other) = 0;
    virtual size_t seq_words(const uint8_t* s, size_t n, int mode, uint64_t* lo, uint64_t* hi, size_t cap) = 0;
    virtual void insert_words(const uint64_t* lo, const uint64_t* hi, size_t n) = 0;
    virtual bool insert_kmer(u128 k) = 0;
    virtual bool contains_kmer(u128 k) const = 0;
    virtual bool contains_word(u128 w) const = 0;
    virtual size_t iter_words(uint64_t* lo, uint64_t* hi, size_t cap) const = 0;
    virtual u128 word_of_kmer(u128 k) const = 0;
    virtual u128 kmer_of_word(u128 w) const = 0;
    virtual size_t n_buckets() const = 0;
    virtual int tid() const = 0;
};

template <class T, class S, int TID> struct Impl : ICBL {
    CBL<T, S> c;
    Impl(unsigned K, unsigned PB, bool canon) : c(K, PB, canon) {}
    void insert_seq(const uint8_t* s, size_t n) override { c.insert_seq(s, n); }
    size_t count() const override { return c.count(); }
    void serialize(Writer& w) const override { c.serialize(w); }
    void load(const uint8_t* d, size_t n) override { c.deserialize(d, n); }
    void merge(ICBL* other) override {
        if (other->tid() != TID) throw std::runtime_error("merge: parameter mismatch");
        c.bitor_assign(static_cast<Impl*>(other)->c);
    }
    size_t seq_words(const uint8_t* s, size_t n, int mode, uint64_t* lo, uint64_t* hi, size_t cap) override {
        if (n < c.P.K) throw std::runtime_error("Sequence size is smaller than K");
        size_t w = 0;
        std::vector<T> tmp;
        c.for_each_chunk(s, n, [&](const uint8_t* ch, size_t m) {
            const std::vector<T>* src;
            if (mode == 0) { c.get_seq_words(ch, m); src = &c.words; }
            else { c.get_seq_words_bruteforce(ch, m, tmp); src = &tmp; }
            for (T x : *src) {
                if (w < cap) { lo[w] = (uint64_t)x; if (hi) hi[w] = (uint64_t)((u128)x >> 64); }
                ++w;
            }
        });
        return w;
    }
    void insert_words(const uint64_t* lo, const uint64_t* hi, size_t n) override {  // WordSet::insert_batch, src/wordset/mod.rs:187-216
        std::vector<T> w(n);
        for (size_t i = 0; i < n; ++i) w[i] = (T)(((u128)(hi ? hi[i] : 0) << 64) | lo[i]);
        c.wordset.insert_batch(w.data(), n);
    }
    bool insert_kmer(u128 k) override { return c.insert_kmer((T)k); }
    bool contains_kmer(u128 k) const override { return c.contains_kmer((T)k); }
    bool contains_word(u128 w) const override { return c.wordset.contains((T)w); }
    u128 word_of_kmer(u128 k) const override { return (u128)c.get_word_bruteforce((T)k); }
    u128 kmer_of_word(u128 w) const override { return (u128)c.recover_kmer((T)w); }
    size_t n_buckets() const override { return c.wordset.tiered.len(); }
    size_t iter_words(uint64_t* lo, uint64_t* hi, size_t cap) const override {  // src/wordset/mod.rs:322-362
        size_t w = 0, rank = 0;
        c.wordset.prefixes.for_each_set([&](size_t prefix) {
            const auto& b = c.wordset.suffix_containers[c.wordset.tiered.get(rank++)];
            for (S s : b.iter_stored(c.P.BYTES)) {
                u128 word = ((u128)prefix << c.P.SUFFIX_BITS) | (u128)s;
                if (w < cap) { lo[w] = (uint64_t)word; if (hi) hi[w] = (uint64_t)(word >> 64); }
                ++w;
            }
        });
        return w;
    }
    int tid() const override { return TID; }
};

ICBL* make(unsigned K, unsigned PB, bool canon) {
    Params p(K, PB);
    if (K < 5 || K > 59) throw std::runtime_error("K must be in [5, 59]");
    if (PB < 1 || PB > 32 || p.SUFFIX_BITS < 1) throw std::runtime_error("PREFIX_BITS must be in [1, min(32, 2K + POS_BITS))");  // library bound (src/wordset/mod.rs:37-41); the CLI bound PB < 2K is build.rs:48-52
    if (p.WORD_BITS <= 64) {
        if (p.BYTES <= 4) return new Impl<uint64_t, uint32_t, 0>(K, PB, canon);
        return new Impl<uint64_t, uint64_t, 1>(K, PB, canon);
    }
    if (p.BYTES <= 8) return new Impl<u128, uint64_t, 2>(K, PB, canon);
    return new Impl<u128, u128, 3>(K, PB, canon);
}

template <class F> int guard(F f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
    catch (...) { g_err = "unknown error"; return 1; }
}
inline u128 mk(uint64_t lo, uint64_t hi) { return ((u128)hi << 64) | lo; }

}  // namespace

extern "C" {

const char* oracle_last_error() { return g_err.c_str(); }

void* oracle_create(unsigned K, unsigned PB, int canonical) {
    ICBL* h = nullptr;
    if (guard([&] { h = make(K, PB, canonical != 0); })) return nullptr;
    return h;
}
void oracle_destroy(void* h) { delete static_cast<ICBL*>(h); }
int oracle_insert_seq(void* h, const uint8_t* seq, uint64_t len) {
    return guard([&] { static_cast<ICBL*>(h)->insert_seq(seq, len); });
}
// One insert_seq call per read, exactly like examples/cbl.rs:160-163. Returns elapsed seconds in *secs.
int oracle_insert_seqs(void* h, const uint8_t* bases, const uint64_t* offsets, uint64_t n, double* secs) {
    return guard([&] {
        auto t0 = std::chrono::steady_clock::now();
        for (uint64_t i = 0; i < n; ++i) static_cast<ICBL*>(h)->insert_seq(bases + offsets[i], offsets[i + 1] - offsets[i]);
        auto t1 = std::chrono::steady_clock::now();
        if (secs) *secs = std::chrono::duration<double>(t1 - t0).count();
    });
}
int oracle_insert_words(void* h, const uint64_t* lo, const uint64_t* hi, uint64_t n) {
    return guard([&] { static_cast<ICBL*>(h)->insert_words(lo, hi, n); });
}
uint64_t oracle_count(void* h) { return static_cast<ICBL*>(h)->count(); }
uint64_t oracle_n_buckets(void* h) { return static_cast<ICBL*>(h)->n_buckets(); }
int oracle_serialize(void* h, uint8_t** buf, uint64_t* len) {
    return guard([&] {
        Writer w;
        static_cast<ICBL*>(h)->serialize(w);
        *buf = (uint8_t*)malloc(w.out.size() ? w.out.size() : 1);
        memcpy(*buf, w.out.data(), w.out.size());
        *len = w.out.size();
    });
}
void oracle_free(void* p) { free(p); }
int oracle_load(void* h, const uint8_t* data, uint64_t len) {
    return guard([&] { static_cast<ICBL*>(h)->load(data, len); });
}
int oracle_merge(void* h, void* other) {
    return guard([&] { static_cast<ICBL*>(h)->merge(static_cast<ICBL*>(other)); });
}
// mode 0: streaming NecklaceQueue path (src/cbl.rs:247-289); mode 1: brute-force necklace_pos per k-mer.
int64_t oracle_seq_words(void* h, const uint8_t* seq, uint64_t len, int mode, uint64_t* lo, uint64_t* hi, uint64_t cap) {
    int64_t n = -1;
    if (guard([&] { n = (int64_t)static_cast<ICBL*>(h)->seq_words(seq, len, mode, lo, hi, cap); })) return -1;
    return n;
}
int oracle_insert_kmer(void* h, uint64_t lo, uint64_t hi) { return static_cast<ICBL*>(h)->insert_kmer(mk(lo, hi)) ? 1 : 0; }
int oracle_contains_kmer(void* h, uint64_t lo, uint64_t hi) { return static_cast<ICBL*>(h)->contains_kmer(mk(lo, hi)) ? 1 : 0; }
int oracle_contains_word(void* h, uint64_t lo, uint64_t hi) { return static_cast<ICBL*>(h)->contains_word(mk(lo, hi)) ? 1 : 0; }
void oracle_word_of_kmer(void* h, uint64_t lo, uint64_t hi, uint64_t* olo, uint64_t* ohi) {
    u128 w = static_cast<ICBL*>(h)->word_of_kmer(mk(lo, hi));
    *olo = (uint64_t)w; *ohi = (uint64_t)(w >> 64);
}
void oracle_kmer_of_word(void* h, uint64_t lo, uint64_t hi, uint64_t* olo, uint64_t* ohi) {
    u128 w = static_cast<ICBL*>(h)->kmer_of_word(mk(lo, hi));
    *olo = (uint64_t)w; *ohi = (uint64_t)(w >> 64);
}
uint64_t oracle_iter_words(void* h, uint64_t* lo, uint64_t* hi, uint64_t cap) { return static_cast<ICBL*>(h)->iter_words(lo, hi, cap); }

// ---- primitives exposed for the reference's KATs -------------------------------------------------------
void oracle_necklace_pos(uint64_t lo, uint64_t hi, unsigned bits, uint64_t* nlo, uint64_t* nhi, unsigned* pos) {
    NeckPos<u128> r = necklace_pos<u128>(mk(lo, hi), bits);
    *nlo = (uint64_t)r.necklace; *nhi = (uint64_t)(r.necklace >> 64); *pos = r.pos;
}
void oracle_revert_necklace_pos(uint64_t lo, uint64_t hi, unsigned pos, unsigned bits, uint64_t* olo, uint64_t* ohi) {
    u128 r = revert_necklace_pos<u128>(mk(lo, hi), pos, bits);
    *olo = (uint64_t)r; *ohi = (uint64_t)(r >> 64);
}
void oracle_rev_comp(uint64_t lo, uint64_t hi, unsigned K, uint64_t* olo, uint64_t* ohi) {
    u128 r = rev_comp<u128>(mk(lo, hi), K);
    *olo = (uint64_t)r; *ohi = (uint64_t)(r >> 64);
}
int oracle_nuc_code(unsigned b) { return nuc_code((uint8_t)b); }

struct QueueH { NecklaceQueue<u128, false> f; NecklaceQueue<u128, true> r; bool rev; QueueH(unsigned b, unsigned w, bool rv) : f(b, w), r(b, w), rev(rv) {} };
void* oracle_queue_new(unsigned bits, unsigned width, int reverse, uint64_t lo, uint64_t hi) {
    QueueH* q = nullptr;
    if (guard([&] { q = new QueueH(bits, width, reverse != 0); if (reverse) q->r.insert_full(mk(lo, hi)); else q->f.insert_full(mk(lo, hi)); })) return nullptr;
    return q;
}
void oracle_queue_free(void* q) { delete static_cast<QueueH*>(q); }
void oracle_queue_insert(void* q, unsigned bit) { QueueH* h = static_cast<QueueH*>(q); if (h->rev) h->r.insert(bit); else h->f.insert(bit); }
void oracle_queue_insert2(void* q, unsigned two) { QueueH* h = static_cast<QueueH*>(q); if (h->rev) h->r.insert2(two); else h->f.insert2(two); }
void oracle_queue_get(void* q, uint64_t* nlo, uint64_t* nhi, unsigned* pos) {
    QueueH* h = static_cast<QueueH*>(q);
    NeckPos<u128> r = h->rev ? h->r.get_necklace_pos() : h->f.get_necklace_pos();
    *nlo = (uint64_t)r.necklace; *nhi = (uint64_t)(r.necklace >> 64); *pos = r.pos;
}

void* oracle_lmq_new(unsigned width) { return new LexMinQueue(width); }
void oracle_lmq_free(void* q) { delete static_cast<LexMinQueue*>(q); }
void oracle_lmq_insert(void* q, uint32_t u) { static_cast<LexMinQueue*>(q)->insert(u); }
void oracle_lmq_insert2(void* q, uint32_t u, uint32_t v) { static_cast<LexMinQueue*>(q)->insert2(u, v); }
void oracle_lmq_insert_full(void* q, const uint32_t* vals) { static_cast<LexMinQueue*>(q)->insert_full([&](unsigned p) { return vals[p]; }); }
unsigned oracle_lmq_min_pos(void* q, uint32_t* out, unsigned cap) {
    unsigned n = 0;
    static_cast<LexMinQueue*>(q)->for_each_min_pos([&](unsigned p) { if (n < cap) out[n] = p; ++n; });
    return n;
}

// dynamic rank bitvector (src/bitvector/mod.rs:148-187) and tiered vector (src/ffi.rs:29-39) KAT hooks
void* oracle_bv_new(uint64_t nbits) { return new FenwickBV(nbits); }
void oracle_bv_free(void* b) { delete static_cast<FenwickBV*>(b); }
int oracle_bv_insert(void* b, uint64_t i) { return !static_cast<FenwickBV*>(b)->set(i); }  // Bitvector::insert = !set
int oracle_bv_contains(void* b, uint64_t i) { return static_cast<FenwickBV*>(b)->get(i); }
uint64_t oracle_bv_rank(void* b, uint64_t i) { return static_cast<FenwickBV*>(b)->rank(i); }
uint64_t oracle_bv_iter(void* b, uint64_t* out, uint64_t cap) {
    uint64_t n = 0;
    static_cast<FenwickBV*>(b)->for_each_set([&](size_t p) { if (n < cap) out[n] = p; ++n; });
    return n;
}
void* oracle_tv_new() { return new TieredVec32(); }
void oracle_tv_free(void* t) { delete static_cast<TieredVec32*>(t); }
void oracle_tv_insert(void* t, uint64_t i, uint32_t v) { static_cast<TieredVec32*>(t)->insert(i, v); }
uint32_t oracle_tv_get(void* t, uint64_t i) { return static_cast<TieredVec32*>(t)->get(i); }
uint64_t oracle_tv_len(void* t) { return static_cast<TieredVec32*>(t)->len(); }

// byte trie KAT hook (src/trie.rs:227-261): insert BYTES-byte big-endian strings, iterate in order
void* oracle_trievec_new() { return new TrieVec<u128>(); }
void oracle_trievec_free(void* t) { delete static_cast<TrieVec<u128>*>(t); }
int oracle_trievec_insert(void* t, uint64_t lo, uint64_t hi, unsigned bytes) { return static_cast<TrieVec<u128>*>(t)->insert(mk(lo, hi), bytes); }
void oracle_trievec_as_trie(void* t, unsigned bytes) { static_cast<TrieVec<u128>*>(t)->as_trie(bytes); }
int oracle_trievec_contains(void* t, uint64_t lo, uint64_t hi, unsigned bytes) { return static_cast<TrieVec<u128>*>(t)->contains(mk(lo, hi), bytes); }
uint64_t oracle_trievec_iter(void* t, unsigned bytes, uint64_t* lo, uint64_t* hi, uint64_t cap) {
    auto v = static_cast<TrieVec<u128>*>(t)->iter_stored(bytes);
    for (size_t i = 0; i < v.size() && i < cap; ++i) { lo[i] = (uint64_t)v[i]; if (hi) hi[i] = (uint64_t)(v[i] >> 64); }
    return v.size();
}

// generic WordSet<PREFIX_BITS, SUFFIX_BITS> KAT hook (src/wordset/mod.rs:451-533)
typedef WordSet<uint64_t, uint64_t> WS64;
void* oracle_ws_new(unsigned pb, unsigned sb) {
    Params p;
    p.PREFIX_BITS = pb; p.SUFFIX_BITS = sb; p.BYTES = (sb + 7) / 8; p.WORD_BITS = pb + sb;
    WS64* w = nullptr;
    if (guard([&] { w = new WS64(p); })) return nullptr;
    return w;
}
void oracle_ws_free(void* w) { delete static_cast<WS64*>(w); }
int oracle_ws_insert(void* w, uint64_t word) { return static_cast<WS64*>(w)->insert(word); }
void oracle_ws_insert_batch(void* w, const uint64_t* words, uint64_t n) { static_cast<WS64*>(w)->insert_batch(words, n); }
int oracle_ws_contains(void* w, uint64_t word) { return static_cast<WS64*>(w)->contains(word); }
uint64_t oracle_ws_count(void* w) { return static_cast<WS64*>(w)->count(); }
uint64_t oracle_ws_iter(void* w, uint64_t* out, uint64_t cap) {
    WS64* s = static_cast<WS64*>(w);
    uint64_t n = 0; size_t rank = 0;
    s->prefixes.for_each_set([&](size_t prefix) {
        for (uint64_t x : s->suffix_containers[s->tiered.get(rank++)].iter_stored(s->P.BYTES)) {
            if (n < cap) out[n] = ((uint64_t)prefix << s->P.SUFFIX_BITS) | x;
            ++n;
        }
    });
    return n;
}

}  // extern "C"
