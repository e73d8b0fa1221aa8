// comm.hpp — the multi-GPU build behind the C ABI: a communicator (RCCL over xGMI, loaded at run time, or host callbacks
// supplied by the embedding program) and the sharded insert that runs on it. Included by cblx.cpp only.
//
// No reference counterpart (the reference is one process); the path is BASELINE.json's north_star: buckets are
// independent by prefix, so the 2^PREFIX_BITS space is cut into `world` contiguous ranges and what the k-mers turn into
// crosses the links once. Same protocol as cbl_amd/sharded.py's "sorted" one (which drives the same device steps through
// torch.distributed): per slice of the rank's reads KRN-1 + the full stable partition, per destination a slice of the
// prefix-sorted batch (prefixes, counts, packed suffixes), one grouped personalised exchange, and at the end one
// bucket-by-bucket merge of the received batches in (slice, source rank) order = stream order.
#pragma once
#include <dlfcn.h>

#include "shard.hpp"

namespace {

// ---- what the sharded build needs from the wire --------------------------------------------------------------------
struct Transport {
    u32 rank = 0, world = 1;
    u64 sent_bytes = 0, recv_bytes = 0, messages = 0;
    virtual ~Transport() {}
    virtual void all_reduce_sum_u64(u64* host_vals, size_t n) = 0;                          // in place, every rank gets the sums
    virtual void all_to_all_u64(const u64* send, u64* recv, size_t per_rank) = 0;         // per_rank values to / from every rank
    // personalised exchange of byte runs in device memory: src holds the runs for rank 0..W-1 back to back (send_off[W + 1]),
    // dst receives the runs of source rank 0..W-1 back to back (recv_off[W + 1]). `after`: stream whose work produced src.
    // Returns once the exchange is ISSUED; wait() returns when everything issued so far has landed.
    virtual void exchange(const u8* d_src, const u64* send_off, u8* d_dst, const u64* recv_off, hipStream_t after) = 0;
    virtual void wait() = 0;
};

// ---- RCCL, resolved at run time (librccl.so.1; a process that already loaded one — torch's — gets that one) ------------
struct Id128 { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* ncclUniqueId by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
enum { RCCL_UINT8 = 1, RCCL_UINT64 = 5, RCCL_SUM = 0 };  // ncclUint8, ncclUint64, ncclSum (rccl.h)
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        auto sym = [&](const char* n) { return dlsym(api.lib, n); };
        api.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(void**, int, Id128, int))sym("ncclCommInitRank");
        api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    });
    if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
        throw Error(CBLX_EDEVICE, "librccl.so.1 is not available (multi-GPU builds need RCCL)");
    return api;
}
#define CBLX_RCCL(expr)                                                                                               \
    do {                                                                                                              \
        const int _r = (expr);                                                                                        \
        if (_r != 0) throw Error(CBLX_EDEVICE, std::string("RCCL error: ") + (rccl().GetErrorString ? rccl().GetErrorString(_r) : "?") + " at " #expr); \
    } while (0)

struct RcclTransport : Transport {
    static constexpr size_t MAX_MSG = 1ull << 30;  // bytes per send / recv call
    void* comm = nullptr;
    int device = 0;
    hipStream_t cs = nullptr;      // the exchange runs on its own stream, next to the kernels of the following slice
    hipEvent_t ev = nullptr;
    u64* d_small = nullptr;        // staging for the small collectives
    size_t small_cap = 0;
    RcclTransport(const u8* id, u32 r, u32 w, int dev) {
        rank = r; world = w; device = dev;
        CBLX_HIP(hipSetDevice(dev));
        Id128 uid;
        std::memcpy(uid.internal, id, sizeof uid.internal);
        CBLX_RCCL(rccl().CommInitRank(&comm, (int)w, uid, (int)r));
        try {
            CBLX_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            CBLX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        } catch (...) {  // a constructor that throws does not run the destructor
            if (cs) (void)hipStreamDestroy(cs);
            (void)rccl().CommDestroy(comm);
            throw;
        }
    }
    ~RcclTransport() override {
        (void)hipSetDevice(device);
        if (cs) (void)hipStreamSynchronize(cs);
        if (comm) (void)rccl().CommDestroy(comm);
        if (d_small) (void)hipFree(d_small);
        if (ev) (void)hipEventDestroy(ev);
        if (cs) (void)hipStreamDestroy(cs);
    }
    u64* small(size_t n) {
        if (small_cap < n) {
            if (d_small) { u64* old = d_small; d_small = nullptr; small_cap = 0; CBLX_HIP(hipFree(old)); }
            const size_t cap = std::max<size_t>(n, 1u << 16);
            CBLX_HIP(hipMalloc((void**)&d_small, cap * 8));
            small_cap = cap;
        }
        return d_small;
    }
    void all_reduce_sum_u64(u64* v, size_t n) override {
        u64* d = small(n);
        CBLX_HIP(hipMemcpyAsync(d, v, n * 8, hipMemcpyHostToDevice, cs));
        CBLX_RCCL(rccl().AllReduce(d, d, n, RCCL_UINT64, RCCL_SUM, comm, cs));
        CBLX_HIP(hipMemcpyAsync(v, d, n * 8, hipMemcpyDeviceToHost, cs));
        CBLX_HIP(hipStreamSynchronize(cs));
    }
    void all_to_all_u64(const u64* send, u64* recv, size_t per) override {
        u64* d = small(2 * per * world);
        u64* ds = d;
        u64* dr = d + per * world;
        CBLX_HIP(hipMemcpyAsync(ds, send, per * world * 8, hipMemcpyHostToDevice, cs));
        CBLX_RCCL(rccl().GroupStart());
        for (u32 p = 0; p < world; ++p) {
            CBLX_RCCL(rccl().Send(ds + p * per, per, RCCL_UINT64, (int)p, comm, cs));
            CBLX_RCCL(rccl().Recv(dr + p * per, per, RCCL_UINT64, (int)p, comm, cs));
        }
        CBLX_RCCL(rccl().GroupEnd());
        CBLX_HIP(hipMemcpyAsync(recv, dr, per * world * 8, hipMemcpyDeviceToHost, cs));
        CBLX_HIP(hipStreamSynchronize(cs));
    }
    void exchange(const u8* src, const u64* so, u8* dst, const u64* ro, hipStream_t after) override {
        CBLX_HIP(hipEventRecord(ev, after));
        CBLX_HIP(hipStreamWaitEvent(cs, ev, 0));
        const u32 me = rank;
        if (so[me + 1] > so[me]) CBLX_HIP(hipMemcpyAsync(dst + ro[me], src + so[me], so[me + 1] - so[me], hipMemcpyDeviceToDevice, cs));
        CBLX_RCCL(rccl().GroupStart());
        for (u32 d = 1; d < world; ++d) {  // ring order keeps the pairing of sends and receives symmetric across ranks
            const u32 to = (me + d) % world, from = (me + world - d) % world;
            for (u64 o = so[to]; o < so[to + 1]; o += MAX_MSG) { CBLX_RCCL(rccl().Send(src + o, (size_t)std::min<u64>(MAX_MSG, so[to + 1] - o), RCCL_UINT8, (int)to, comm, cs)); ++messages; }
            for (u64 o = ro[from]; o < ro[from + 1]; o += MAX_MSG) { CBLX_RCCL(rccl().Recv(dst + o, (size_t)std::min<u64>(MAX_MSG, ro[from + 1] - o), RCCL_UINT8, (int)from, comm, cs)); ++messages; }
        }
        CBLX_RCCL(rccl().GroupEnd());
        sent_bytes += so[world] - (so[me + 1] - so[me]);
        recv_bytes += ro[world] - (ro[me + 1] - ro[me]);
    }
    void wait() override { CBLX_HIP(hipStreamSynchronize(cs)); }
};

// ---- host callbacks (include/cblx.h: cblx_transport): the embedding program moves the bytes -----------------------------
struct CallbackTransport : Transport {
    cblx_transport t;
    CallbackTransport(const cblx_transport& tt, u32 r, u32 w) : t(tt) { rank = r; world = w; }
    void chk(int rc, const char* what) { if (rc != 0) throw Error(CBLX_EDEVICE, std::string("transport callback failed: ") + what); }
    void all_reduce_sum_u64(u64* v, size_t n) override { chk(t.all_reduce_sum_u64(t.user, v, n), "all_reduce_sum_u64"); }
    void all_to_all_u64(const u64* s, u64* r, size_t per) override { chk(t.all_to_all_u64(t.user, s, r, per), "all_to_all_u64"); }
    void exchange(const u8* src, const u64* so, u8* dst, const u64* ro, hipStream_t after) override {
        CBLX_HIP(hipStreamSynchronize(after));  // the callback sees finished data and is complete on return
        chk(t.exchange(t.user, src, so, dst, ro), "exchange");
        sent_bytes += so[world] - (so[rank + 1] - so[rank]);
        recv_bytes += ro[world] - (ro[rank + 1] - ro[rank]);
        messages += 2 * (world - 1);
    }
    void wait() override {}
};

}  // namespace

struct cblx_comm {
    std::unique_ptr<Transport> t;
    int device = 0;
    std::string err;
};

namespace {

// ---- splitters: quantiles of a sampled prefix histogram (necklace prefixes are heavily skewed) -----------
static const u32 SPLIT_HIST_BITS = 16, SPLIT_STRIDE = 61;
template <typename HiT>
__global__ void k_sample_hist(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n, u32 stride, u32 shift, u32 nbits, u32* __restrict__ hist) {
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i >= n) return;
    atomicAdd(&hist[get_bits(lo[i], (u64)ld_hi<HiT>(hi, i), shift, nbits)], 1u);
}
// world-1 ascending prefix values cutting the histogram mass into `world` near-equal parts (same rule as cbl_amd/sharded.py)
std::vector<u32> choose_bounds(const std::vector<u64>& hist, u32 world, u32 PB, u32 hb) {
    std::vector<u32> b;
    const size_t nh = hist.size();
    std::vector<double> cum(nh);
    double run = 0;
    for (size_t i = 0; i < nh; ++i) { run += (double)hist[i]; cum[i] = run; }
    const double total = run;
    const int shift = (int)PB - (int)hb;
    for (u32 d = 1; d < world; ++d) {
        u64 cell;
        if (total > 0) cell = (u64)(std::lower_bound(cum.begin(), cum.end(), total * d / world) - cum.begin()) + 1;
        else cell = (u64)d * nh / world;
        cell = std::min<u64>(std::max<u64>(cell, 1), nh);
        const u64 v = shift >= 0 ? std::min<u64>(cell << shift, (1ull << PB) - 1) : cell >> -shift;
        b.push_back((u32)v);
    }
    for (size_t i = 1; i < b.size(); ++i) b[i] = std::max(b[i], b[i - 1]);
    return b;
}

template <typename C>
void sharded_insert(cblx_ctx* c, Transport& T, const u8* d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, u32* bounds, int* bounds_valid) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    const u32 W = T.world, B = P.BYTES;
    struct Slice {
        Buf<u32> sp, sc, rp, rc;   // sent / received prefixes and counts
        Buf<u8> ss, rs;            // sent / received packed suffixes
        std::vector<u64> rb, rw;   // received buckets / words per source rank
    };
    std::vector<Slice> sl(nslices);
    // an error between two slices must not hand the buffers of an exchange that is still running back to the pool
    struct Drain { Transport& t; ~Drain() { try { t.wait(); } catch (...) {} } } drain{T};
    for (u32 s = 0; s < nslices; ++s) {
        const u64 a = cuts[s], b = cuts[s + 1];
        if (b < a || b > n) throw Error(CBLX_EINVAL, "slice cuts must be ascending and at most n");
        if (!*bounds_valid) {
            // first batch only: quantile ranges from the all-reduced, sampled prefix histogram of this slice's words
            const u32 hb = std::min(SPLIT_HIST_BITS, P.PB);
            std::vector<u64> hist((size_t)1 << hb, 0);
            if (b > a) {
                ChunkPlan pl;
                const u8* pb = d_bases;
                plan_chunks(c, pb, d_offsets + a, b - a, pl);
                if (pl.n_kmers) {
                    Buf<u64> w_lo(c->pool, pl.n_kmers + 2);
                    Buf<u8> w_hi(c->pool, (pl.n_kmers + 2) * std::max<size_t>(1, hi_elem_size(P)));
                    Buf<u32> d_hist(c->pool, (size_t)1 << hb);
                    CBLX_HIP(hipMemsetAsync(d_hist.get(), 0, ((size_t)1 << hb) * 4, c->stream));
                    encode<C>(c, pb, pl, w_lo.get(), (HiT*)w_hi.get(), 0);
                    hipLaunchKernelGGL(k_sample_hist<HiT>, grid1(ceil_div(pl.n_kmers, SPLIT_STRIDE), 256), dim3(256), 0, c->stream, w_lo.get(), (const HiT*)w_hi.get(), pl.n_kmers,
                                       SPLIT_STRIDE, P.SB + P.PB - hb, hb, d_hist.get());
                    CBLX_HIP(hipGetLastError());
                    std::vector<u32> h32 = d2h_vec<u32>(c, d_hist.get(), (size_t)1 << hb);
                    for (size_t i = 0; i < h32.size(); ++i) hist[i] = h32[i];
                }
            }
            T.all_reduce_sum_u64(hist.data(), hist.size());
            const std::vector<u32> bb = choose_bounds(hist, W, P.PB, hb);
            for (u32 d = 0; d + 1 < W; ++d) bounds[d] = bb[d];
            *bounds_valid = 1;
        }
        std::vector<u64> bs(W + 1), ws(W + 1);
        sorted_batch_begin<C>(c, d_bases, d_offsets + a, b - a, bounds, W, bs.data(), ws.data());
        Slice& S = sl[s];
        S.sp = Buf<u32>(c->pool, bs[W] + 1);
        S.sc = Buf<u32>(c->pool, bs[W] + 1);
        S.ss = Buf<u8>(c->pool, ws[W] * B + 16);
        sorted_batch_export<C>(c, S.sp.get(), S.sc.get(), S.ss.get());
        std::vector<u64> send(2 * W), recv(2 * W);
        for (u32 d = 0; d < W; ++d) { send[2 * d] = bs[d + 1] - bs[d]; send[2 * d + 1] = ws[d + 1] - ws[d]; }
        T.all_to_all_u64(send.data(), recv.data(), 2);
        S.rb.resize(W); S.rw.resize(W);
        std::vector<u64> so4(W + 1), ro4(W + 1), soB(W + 1), roB(W + 1);
        so4[0] = ro4[0] = soB[0] = roB[0] = 0;
        for (u32 d = 0; d < W; ++d) {
            S.rb[d] = recv[2 * d]; S.rw[d] = recv[2 * d + 1];
            so4[d + 1] = so4[d] + send[2 * d] * 4;       ro4[d + 1] = ro4[d] + S.rb[d] * 4;
            soB[d + 1] = soB[d] + send[2 * d + 1] * B;   roB[d + 1] = roB[d] + S.rw[d] * B;
        }
        S.rp = Buf<u32>(c->pool, ro4[W] / 4 + 1);
        S.rc = Buf<u32>(c->pool, ro4[W] / 4 + 1);
        S.rs = Buf<u8>(c->pool, roB[W] + 16);
        T.exchange((const u8*)S.sp.get(), so4.data(), (u8*)S.rp.get(), ro4.data(), c->stream);
        T.exchange((const u8*)S.sc.get(), so4.data(), (u8*)S.rc.get(), ro4.data(), c->stream);
        T.exchange(S.ss.get(), soB.data(), S.rs.get(), roB.data(), c->stream);
    }
    T.wait();
    std::vector<cblx_batch_view> views;  // stream order: slice-major, source-rank-minor
    for (u32 s = 0; s < nslices; ++s) {
        Slice& S = sl[s];
        u64 bo = 0, wo = 0;
        for (u32 r = 0; r < W; ++r) {
            if (S.rw[r]) views.push_back(cblx_batch_view{S.rb[r], S.rw[r], S.rp.get() + bo, S.rc.get() + bo, S.rs.get() + wo * B});
            bo += S.rb[r];
            wo += S.rw[r];
        }
        S.sp.reset(); S.sc.reset(); S.ss.reset();  // sent data is done with
    }
    if (!views.empty()) insert_sorted_batches<C>(c, views.data(), (u32)views.size());
    CBLX_HIP(hipStreamSynchronize(c->stream));
}

}  // namespace
