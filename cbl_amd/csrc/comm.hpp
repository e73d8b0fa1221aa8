// comm.hpp — the multi-GPU build behind the C ABI: a communicator (RCCL over xGMI, loaded at run time, or host callbacks
// supplied by the embedding program) and the sharded insert that runs on it. Included by cblx.cpp only.
//
// No reference counterpart (the reference is one process); the path is BASELINE.json's north_star: buckets are
// independent by prefix, so the 2^PREFIX_BITS space is cut into `world` contiguous ranges (quantiles of a sampled, all-reduced
// prefix histogram) and what the k-mers turn into crosses the links once. Three protocols, same result (byte-identical to the
// one-process build in the job's stream order, slice-major / rank-minor):
//   "bins", grouped receiver (default on an empty index; sharded_insert_grouped): the senders run KRN-1 + the FIRST partition pass on
//     bins that refine that pass's digit by the destination rank and by G groups per rank; 8-byte records + a digit byte cross the
//     links group-major, and the receiver runs the remaining passes, its window of the directory and the bucket kernels of group g
//     while groups g+1.. are still on the wire;
//   "bins", plain receiver (sharded_insert_bins: non-empty index, degenerate bounds, one rank, the sliced one-GPU host input): the
//     same records slice-major, everything behind the first pass once the last record has landed;
//   "sorted" (sharded_insert_sorted): the sender partitions completely and ships prefixes, counts and packed suffixes; the receiver
//     merges the batches run by run — a third fewer bytes on the wire, one more pass over the words (2 - 4 GPUs, PREFIX_BITS <= 8).
// Transports: RCCL (grouped ncclSend / ncclRecv on a side stream), host callbacks (tests: ranks sharing one GPU), the one-rank
// local transport (a batch arriving over PCIe), and the recording / replaying pair of the one-GPU rehearsal (SimTransport).
#pragma once
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <atomic>
#include <map>
#include <functional>

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
    // Several personalised exchanges issued as ONE (grouped receiver: a group's share of every slice and of every record array). The
    // runs of an item lie anywhere in its arrays: bytes [s_off[d], s_off[d] + s_len[d]) of src go to rank d, the bytes from rank r
    // land at [r_off[r], r_off[r] + r_len[r]) of dst; the own rank's entries are ignored (its records never leave the device).
    struct Item { const u8* src; u8* dst; std::vector<u64> s_off, s_len, r_off, r_len; };
    virtual void exchange_items(const std::vector<Item>& items, hipStream_t after) = 0;
    // `e` fires once everything issued so far has landed (`after`: the stream of a transport that completes on return)
    virtual void record(hipEvent_t e, hipStream_t after) { CBLX_HIP(hipEventRecord(e, after)); }
    virtual void begin_job() {}  // a sharded insert starts (the rehearsal transports count their calls from here)
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
    void exchange_items(const std::vector<Item>& items, hipStream_t after) override {
        CBLX_HIP(hipEventRecord(ev, after));
        CBLX_HIP(hipStreamWaitEvent(cs, ev, 0));
        const u32 me = rank;
        CBLX_RCCL(rccl().GroupStart());
        for (u32 d = 1; d < world; ++d) {  // ring order, items in the same order on both ends: the k-th send to a peer meets its k-th receive
            const u32 to = (me + d) % world, from = (me + world - d) % world;
            for (const Item& it : items) {
                for (u64 o = 0; o < it.s_len[to]; o += MAX_MSG) { CBLX_RCCL(rccl().Send(it.src + it.s_off[to] + o, (size_t)std::min<u64>(MAX_MSG, it.s_len[to] - o), RCCL_UINT8, (int)to, comm, cs)); ++messages; }
                for (u64 o = 0; o < it.r_len[from]; o += MAX_MSG) { CBLX_RCCL(rccl().Recv(it.dst + it.r_off[from] + o, (size_t)std::min<u64>(MAX_MSG, it.r_len[from] - o), RCCL_UINT8, (int)from, comm, cs)); ++messages; }
            }
        }
        CBLX_RCCL(rccl().GroupEnd());
        for (const Item& it : items)
            for (u32 r = 0; r < world; ++r) if (r != me) { sent_bytes += it.s_len[r]; recv_bytes += it.r_len[r]; }
    }
    void record(hipEvent_t e, hipStream_t) override { CBLX_HIP(hipEventRecord(e, cs)); }
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
    // the callback moves back-to-back runs: an item's runs are packed into a staging buffer, exchanged, and unpacked (tests only)
    void exchange_items(const std::vector<Item>& items, hipStream_t after) override {
        CBLX_HIP(hipStreamSynchronize(after));
        for (const Item& it : items) {
            std::vector<u64> so(world + 1, 0), ro(world + 1, 0);
            for (u32 r = 0; r < world; ++r) { so[r + 1] = so[r] + (r == rank ? 0 : it.s_len[r]); ro[r + 1] = ro[r] + (r == rank ? 0 : it.r_len[r]); }
            u8 *ts = nullptr, *tr = nullptr;
            CBLX_HIP(hipMalloc((void**)&ts, so[world] + 16));
            if (hipMalloc((void**)&tr, ro[world] + 16) != hipSuccess) { (void)hipFree(ts); throw Error(CBLX_ENOMEM, "staging buffer of the callback transport"); }
            struct Free { u8 *a, *b; ~Free() { (void)hipFree(a); (void)hipFree(b); } } fr{ts, tr};
            for (u32 r = 0; r < world; ++r) if (r != rank && it.s_len[r]) CBLX_HIP(hipMemcpy(ts + so[r], it.src + it.s_off[r], it.s_len[r], hipMemcpyDeviceToDevice));
            CBLX_HIP(hipDeviceSynchronize());
            chk(t.exchange(t.user, ts, so.data(), tr, ro.data()), "exchange");
            for (u32 r = 0; r < world; ++r) if (r != rank && it.r_len[r]) CBLX_HIP(hipMemcpy(it.dst + it.r_off[r], tr + ro[r], it.r_len[r], hipMemcpyDeviceToDevice));
            CBLX_HIP(hipDeviceSynchronize());
            sent_bytes += so[world];
            recv_bytes += ro[world];
            messages += 2 * (world - 1);
        }
    }
};


// ---- rehearsal of ONE rank of a W-GPU job on one GPU (dev / bench: tools/emulate_wire.py; cblx_comm_init_sim) ----------------------
// No multi-GPU node was available while this was built, so the schedule of the sharded insert — what the receiver's kernels hide of
// the wire — is measured on one GPU: ranks 1 .. W-1 run one after the other on a RECORDING transport that keeps what each of them
// would send to rank 0 (headers and device bytes, per call); then rank 0 runs for real on a REPLAYING transport: its collectives are
// answered from the records, and the bytes of every exchange are copied into its receive arena on a side stream that a host function
// holds back until a wire of `link_gbps` per source rank (W-1 links working in parallel, one per peer, as on an xGMI mesh) would
// have delivered them. The copies read and write what RCCL's sends and receives would read and write (HBM traffic of both
// directions); what is not emulated: the CUs RCCL's kernels occupy, and link contention. Sums over ranks are answered as W times the
// own value (the ranks of the rehearsal hold equally many reads); the group cuts travel through the store so that every rank of the
// rehearsal uses rank 1's.
struct SimStore {
    struct Buf8 { u8* p = nullptr; u64 n = 0; };
    u32 world = 0;
    u32 target = 0;  // the rank that replays (CBLX_SIM_TARGET when the store is created; 0 = the densest prefix range, W - 1 = the sparse tail)
    std::vector<std::vector<std::vector<u64>>> a2a;             // [rank][call] -> what the rank sends the target
    std::vector<std::vector<std::vector<Buf8>>> xch;            // [rank][call][item] -> bytes for rank 0
    std::vector<u32> g_bounds, g_cuts;
    bool have_cuts = false;
    ~SimStore() { for (auto& r : xch) for (auto& cl : r) for (auto& b : cl) if (b.p) (void)hipFree(b.p); }
    static std::mutex& mu() { static std::mutex m; return m; }
    // (shared ownership: a communicator keeps its store alive after cblx_sim_store_free has dropped the name)
    static std::map<u64, std::shared_ptr<SimStore>>& all() { static std::map<u64, std::shared_ptr<SimStore>> m; return m; }
    static std::shared_ptr<SimStore> get(u64 id, u32 world) {
        std::lock_guard<std::mutex> g(mu());
        auto it = all().find(id);
        if (it != all().end()) {
            if (it->second->world != world) throw Error(CBLX_EINVAL, "rehearsal store: created for another world size");
            return it->second;
        }
        if (world == 0 || world > CUT_MAX_DEST) throw Error(CBLX_EINVAL, "rehearsal store: world size out of range");
        std::shared_ptr<SimStore> p(new SimStore());
        p->world = world; p->a2a.resize(world); p->xch.resize(world);
        const char* e = std::getenv("CBLX_SIM_TARGET");
        const u32 t = e ? (u32)std::strtoul(e, nullptr, 10) : 0u;
        p->target = t < world ? t : 0u;
        all()[id] = p;
        return p;
    }
};
struct SimTransport : Transport {
    std::shared_ptr<SimStore> st;
    double link_gbps;
    size_t n_a2a = 0, n_x = 0;
    hipStream_t ps = nullptr;  // replay: the "wire"
    hipEvent_t ev = nullptr;
    struct Pace { std::chrono::steady_clock::time_point start, done; bool any = false; } pace;
    struct Gate { SimTransport* t; double seconds; bool opens; };
    std::vector<std::unique_ptr<Gate>> gates;
    SimTransport(std::shared_ptr<SimStore> s, u32 r, u32 w, double gbps) : st(std::move(s)), link_gbps(gbps) {
        rank = r; world = w;
        if (r == st->target) {
            CBLX_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
            CBLX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
    }
    ~SimTransport() override {
        if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
        if (ev) (void)hipEventDestroy(ev);
    }
    void begin_job() override {
        n_a2a = n_x = 0;
        if (rank != st->target) { st->a2a[rank].clear(); for (auto& cl : st->xch[rank]) for (auto& b : cl) if (b.p) (void)hipFree(b.p); st->xch[rank].clear(); }
        else { if (ps) CBLX_HIP(hipStreamSynchronize(ps)); gates.clear(); pace.any = false; }
    }
    void all_reduce_sum_u64(u64* v, size_t n) override { for (size_t i = 0; i < n; ++i) v[i] *= world; }
    void all_to_all_u64(const u64* send, u64* recv, size_t per) override {
        std::memset(recv, 0, per * world * 8);
        std::memcpy(recv + rank * per, send + rank * per, per * 8);
        const u32 tg = st->target;
        if (rank != tg) { st->a2a[rank].push_back(std::vector<u64>(send + (size_t)tg * per, send + (size_t)(tg + 1) * per)); return; }  // (what goes to the target)
        for (u32 r = 0; r < world; ++r) {
            if (r == tg) continue;
            if (st->a2a[r].size() <= n_a2a || st->a2a[r][n_a2a].size() != per) throw Error(CBLX_EINVAL, "rehearsal: rank " + std::to_string(r) + " was not recorded with this schedule");
            std::memcpy(recv + r * per, st->a2a[r][n_a2a].data(), per * 8);
        }
        ++n_a2a;
    }
    static void gate_fn(void* p) {
        Gate* g = (Gate*)p;
        Pace& pc = g->t->pace;
        const auto now = std::chrono::steady_clock::now();
        if (g->opens) { pc.start = pc.any && pc.done > now ? pc.done : now; return; }  // the wire takes the call when it is free
        pc.done = pc.start + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(g->seconds));
        pc.any = true;
        std::this_thread::sleep_until(pc.done);
    }
    void items_common(const std::vector<Item>& items, hipStream_t after) {
        const u32 tg = st->target;
        if (rank != tg) {  // record what goes to the target
            CBLX_HIP(hipStreamSynchronize(after));
            std::vector<SimStore::Buf8> call;
            for (const Item& it : items) {
                SimStore::Buf8 b;
                b.n = it.s_len[tg];
                if (b.n) { CBLX_HIP(hipMalloc((void**)&b.p, b.n)); CBLX_HIP(hipMemcpy(b.p, it.src + it.s_off[tg], b.n, hipMemcpyDeviceToDevice)); }
                call.push_back(b);
                sent_bytes += b.n;
            }
            st->xch[rank].push_back(std::move(call));
            return;
        }
        CBLX_HIP(hipEventRecord(ev, after));
        CBLX_HIP(hipStreamWaitEvent(ps, ev, 0));
        gates.emplace_back(new Gate{this, 0.0, true});
        CBLX_HIP(hipLaunchHostFunc(ps, gate_fn, gates.back().get()));
        u64 worst = 0;
        for (u32 r = 0; r < world; ++r) {
            if (r == tg) continue;
            if (st->xch[r].size() <= n_x || st->xch[r][n_x].size() != items.size()) throw Error(CBLX_EINVAL, "rehearsal: rank " + std::to_string(r) + " was not recorded with this schedule");
            u64 from_r = 0;
            for (size_t i = 0; i < items.size(); ++i) {
                const SimStore::Buf8& b = st->xch[r][n_x][i];
                if (b.n != items[i].r_len[r]) throw Error(CBLX_EDEVICE, "rehearsal: rank " + std::to_string(r) + " recorded " + std::to_string(b.n) + " bytes, rank 0 expects " + std::to_string(items[i].r_len[r]));
                if (b.n) CBLX_HIP(hipMemcpyAsync(items[i].dst + items[i].r_off[r], b.p, b.n, hipMemcpyDeviceToDevice, ps));
                from_r += b.n;
            }
            recv_bytes += from_r;
            worst = std::max(worst, from_r);
        }
        for (const Item& it : items) for (u32 r = 0; r < world; ++r) if (r != tg) sent_bytes += it.s_len[r];
        messages += 2 * (world - 1) * items.size();
        gates.emplace_back(new Gate{this, link_gbps > 0 ? (double)worst / (link_gbps * 1e9) : 0.0, false});
        CBLX_HIP(hipLaunchHostFunc(ps, gate_fn, gates.back().get()));
        ++n_x;
    }
    void exchange_items(const std::vector<Item>& items, hipStream_t after) override { items_common(items, after); }
    void exchange(const u8* src, const u64* so, u8* dst, const u64* ro, hipStream_t after) override {
        Item it{src, dst, {}, {}, {}, {}};
        for (u32 r = 0; r < world; ++r) { it.s_off.push_back(so[r]); it.s_len.push_back(so[r + 1] - so[r]); it.r_off.push_back(ro[r]); it.r_len.push_back(ro[r + 1] - ro[r]); }
        if (so[rank + 1] > so[rank]) CBLX_HIP(hipMemcpyAsync(dst + ro[rank], src + so[rank], so[rank + 1] - so[rank], hipMemcpyDeviceToDevice, after));
        items_common(std::vector<Item>{it}, after);
    }
    void wait() override { if (ps) CBLX_HIP(hipStreamSynchronize(ps)); }
    void record(hipEvent_t e, hipStream_t after) override { CBLX_HIP(hipEventRecord(e, ps ? ps : after)); }
};
}  // namespace

struct cblx_comm {
    std::unique_ptr<Transport> t;
    int device = 0;
    u32 protocol = CBLX_PROTO_AUTO;
    u32 protocol_used = CBLX_PROTO_BINS;  // what the last sharded insert resolved AUTO to
    std::string err;
    // grouped receiver (sharded_insert_grouped): groups per rank asked for (0: CBLX_RECV_GROUPS or the default), the group cuts chosen
    // together with the bounds they refine, and how many groups the last call worked through (0: it took the ungrouped path)
    u32 recv_groups = 0, groups_used = 0;
    u32 groups_fine = 0;  // ... of which sorted 16 prefix bits behind the first pass (FINE bins, PREFIX_BITS > 24)
    std::vector<u32> g_bounds, g_cuts;
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

// the all-reduced, sampled prefix histogram (2^min(16, PB) bins) of one slice's words
template <typename C>
std::vector<u64> sampled_prefix_hist(cblx_ctx* c, Transport& T, const u8* d_bases, const u64* d_offsets, u64 nseq) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    const u32 hb = std::min(SPLIT_HIST_BITS, P.PB);
    std::vector<u64> hist((size_t)1 << hb, 0);
    if (nseq) {
        ChunkPlan pl;
        const u8* pb = d_bases;
        plan_chunks(c, pb, d_offsets, nseq, pl);
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
    return hist;
}
// first batch only: quantile ranges from the all-reduced, sampled prefix histogram of one slice's words
template <typename C>
void choose_bounds_from_slice(cblx_ctx* c, Transport& T, const u8* d_bases, const u64* d_offsets, u64 nseq, u32* bounds, std::vector<u64>* hist_out = nullptr, u32 fine_groups = 0) {
    const Consts& P = c->P;
    const u32 hb = std::min(SPLIT_HIST_BITS, P.PB);
    std::vector<u64> hist = sampled_prefix_hist<C>(c, T, d_bases, d_offsets, nseq);
    if (fine_groups) weigh_tail_for_fine_bins(hist, P.PB, T.world, fine_groups, hb);  // (the "bins" protocol with its grouped receiver: FINE bins at PREFIX_BITS > 24)
    const std::vector<u32> bb = choose_bounds(hist, T.world, P.PB, hb);
    for (u32 d = 0; d + 1 < T.world; ++d) bounds[d] = bb[d];
    if (hist_out) *hist_out = std::move(hist);
}

// ---- protocol "sorted": the sender partitions completely; prefixes, counts and packed suffixes on the wire; the receiver
// merges the batches run by run (fewest bytes per word: the choice when the links are the bound, world <= 4) ----------------
template <typename C>
void sharded_insert_sorted(cblx_ctx* c, Transport& T, const u8* d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, u32* bounds, int* bounds_valid) {
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
            choose_bounds_from_slice<C>(c, T, d_bases, d_offsets + a, b - a, bounds);
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

// ---- protocol "bins": the exchange rides between the first and the second partition pass ------------------------------------
// The sender runs KRN-1 and pass A only — on BINS (DigitBin: the pass-A segment refined by the destination rank), so its output
// is contiguous per destination and, inside a destination, per segment; the records of its own range go straight into its
// receive arena (OwnWindow), the others leave as they are: 8-byte records (the hi byte of a 65..72-bit word is implied by the
// segment, as on one GPU) plus the 1-byte digit side channel of the next pass. The receiver keeps the arena as the input of
// the remaining passes: a piece table (slice, source, segment) -> tile table of the first LSD pass (k_piece_tables), after
// which the one-GPU pipeline continues unchanged. No pass is added to the one-GPU path and no record is copied: per rank the
// job costs what the direct build costs plus the slices' fixed costs.
struct BinMap {
    u32 v_of[256], d_of[256];  // bin -> segment, destination (0xFFFFFFFF: no such bin)
    bool ok = true;
};
inline BinMap make_bin_map(const Consts& P, const u32* bounds, u32 W) {
    BinMap M;
    for (u32 i = 0; i < 256; ++i) M.v_of[i] = M.d_of[i] = 0xFFFFFFFFu;
    const u32 RB = P.PB - 8;
    auto dest = [&](u64 p) { u32 d = 0; for (u32 i = 0; i + 1 < W; ++i) d += bounds[i] <= p ? 1u : 0u; return d; };
    for (u32 i = 0; i + 1 < W; ++i) if ((u64)bounds[i] > (255ull << RB)) M.ok = false;  // the all-ones segment must belong to one rank
    for (u32 v = 0; v < 128; ++v)
        for (u32 d = dest((u64)v << RB); d <= dest((((u64)v + 1) << RB) - 1); ++d) { M.v_of[v + d] = v; M.d_of[v + d] = d; }
    M.v_of[255] = 255; M.d_of[255] = W - 1;
    return M;
}
inline bool bins_protocol_fits(const Consts& P, const u32* bounds, u32 W) { return P.PB >= 9 && make_bin_map(P, bounds, W).ok; }

// `before_slice` (optional): called before slice s is touched — where a caller whose slices are still arriving (a batch on its way
// over PCIe, flush()) makes the ctx's stream wait for slice s; before_slice(~0u) precedes the first read of the offsets.
template <typename C, typename Hook>
void sharded_insert_bins(cblx_ctx* c, Transport& T, const BaseView& d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, const u32* bounds, Hook&& before_slice) {
    typedef typename C::HiT HiT;
    constexpr bool DROP_HI = std::is_same<HiT, u8>::value;
    typedef typename std::conditional<DROP_HI, NoHi, HiT>::type OutH;  // record layout behind pass A
    constexpr size_t OHS = HiTraits<OutH>::has ? sizeof(u64) : 0;       // bytes of the hi part on the wire (u64 or none)
    const Consts& P = c->P;
    const u32 W = T.world, me = T.rank, RB = P.PB - 8;
    const BinMap M = make_bin_map(P, bounds, W);
    const LsdPlan LP = lsd_plan(P, false);  // (the receiver runs LSD passes only: pipeline.hpp)
    const DigitBits nextd{P.SB + LP.sh[0], LP.wid[0]};
    DigitBin fn;
    fn.SB = P.SB; fn.PB = P.PB; fn.RB = RB; fn.nd = W;
    EncHist eh0{};
    eh0.nd = W; eh0.SB = P.SB; eh0.PB = P.PB; eh0.binRB = RB;
    for (u32 i = 0; i < MAX_DEST - 1; ++i) { fn.bounds[i] = i + 1 < W ? bounds[i] : 0xFFFFFFFFu; eh0.bounds[i] = fn.bounds[i]; }

    // receive arena: the slices land back to back; capacity from the job's k-mer count (an upper bound: bases - (K - 1) per
    // sequence), the rank's share of it with slack for uneven ranges; it grows when a slice does not fit
    for (u32 s = 0; s < nslices; ++s) if (cuts[s + 1] < cuts[s] || cuts[s + 1] > n) throw Error(CBLX_EINVAL, "slice cuts must be ascending and at most n");
    const u64 n0 = cuts[0], n1 = cuts[nslices];
    u64 mine = 0;
    before_slice(~0u);
    if (n1 > n0) {
        const u64 first = d2h<u64>(c, d_offsets + n0), last = d2h<u64>(c, d_offsets + n1);
        if (last < first) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
        const u64 sub = (n1 - n0) * (u64)(P.K - 1);
        mine = last - first > sub ? last - first - sub : 0;
    }
    u64 job = mine;
    T.all_reduce_sum_u64(&job, 1);
    const u64 LIMIT = 0xFFFFFFF0ull - 2 * RDX_TILE;  // one round = one batch of the pipeline (32-bit positions)
    const bool trace = std::getenv("CBLX_TRACE_SHARDED") != nullptr;
    u64 cap = W == 1 ? mine : std::min<u64>(job, job / W + job / (4 * W) + (1u << 20));
    cap = std::min<u64>(std::max<u64>(cap, 1024), LIMIT);
    Buf<u64> a_lo;
    Buf<u8> a_hi, a_dig;
    auto alloc_arena = [&](u64 ncap, Buf<u64>& lo, Buf<u8>& hi, Buf<u8>& dg) {
        lo = Buf<u64>(c->pool, ncap + 2);
        hi = Buf<u8>(c->pool, OHS ? (ncap + 2) * OHS : 8);
        dg = Buf<u8>(c->pool, ncap + 64);
    };
    // send buffers of the exchanges in flight. A slice's buffers go back to the pool as soon as its exchange has completed (an event
    // behind it on the transport's stream, polled at the start of every later slice): what stays allocated is bounded by the slices the
    // wire is behind, not by the job (a whole call's send buffers — (W-1)/W of the rank's words, 9 bytes each — next to the receive
    // arena and the pipeline's twin could run one rank out of memory while the others wait in a collective)
    struct Sent { Buf<u64> lo; Buf<u8> hi, dig; hipEvent_t done = nullptr; };
    std::vector<Sent> sent;
    struct SentEvents { std::vector<Sent>& v; ~SentEvents() { for (Sent& x : v) if (x.done) (void)hipEventDestroy(x.done); } } sent_events{sent};
    auto reap_sent = [&]() {
        for (size_t i = 0; i < sent.size();) {
            if (sent[i].done && hipEventQuery(sent[i].done) == hipSuccess) {
                (void)hipEventDestroy(sent[i].done);
                sent[i] = std::move(sent.back());
                sent.pop_back();
            } else { (void)hipGetLastError(); ++i; }
        }
    };
    std::vector<u32> pcnt, pbase;      // piece table of the round: [piece][256] counts, arena position of every piece
    // one rank (a batch arriving over PCIe, insert_device_sliced): everything is the rank's own, nothing has to be agreed per slice —
    // the bin counts of a slice stay on the device until the round ends, and the slices run without a host round trip between
    // KRN-1 and pass A
    struct Deferred { size_t piece; u64 n; Buf<u32> coltot; };
    struct Work { ChunkPlan pl; Buf<u64> t_lo; Buf<u8> t_hi; Buf<u32> counts, colpre, scratch, adj; };  // workspace of one slice
    Work prev_work;
    std::vector<Deferred> deferred;
    u64 filled = 0;
    struct Drain { Transport& t; ~Drain() { try { t.wait(); } catch (...) {} } } drain{T};
    auto grow = [&](u64 need) {
        T.wait();
        CBLX_HIP(hipStreamSynchronize(c->stream));
        const u64 ncap = std::min<u64>(LIMIT, need + need / 4 + 4096);
        Buf<u64> lo;
        Buf<u8> hi, dg;
        alloc_arena(ncap, lo, hi, dg);
        if (filled) {
            CBLX_HIP(hipMemcpyAsync(lo.get(), a_lo.get(), filled * 8, hipMemcpyDeviceToDevice, c->stream));
            if (OHS) CBLX_HIP(hipMemcpyAsync(hi.get(), a_hi.get(), filled * OHS, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(dg.get(), a_dig.get(), filled, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipStreamSynchronize(c->stream));
        }
        a_lo = std::move(lo); a_hi = std::move(hi); a_dig = std::move(dg);
        cap = ncap;
    };
    // what has been received so far goes through the rest of the pipeline (end of the call, or the arena would pass 2^32 records)
    auto finish_round = [&]() {
        T.wait();
        CBLX_HIP(hipStreamSynchronize(c->stream));
        for (Sent& x : sent) if (x.done) { (void)hipEventDestroy(x.done); x.done = nullptr; }
        sent.clear();
        if (trace) fprintf(stderr, "[cblx bins] rank %u round ends: filled=%llu pieces=%zu\n", me, (unsigned long long)filled, pbase.size());
        for (Deferred& d : deferred) {
            const std::vector<u32> t = d2h_vec<u32>(c, d.coltot.get(), 256);
            u64 sum = 0;
            for (u32 bin = 0; bin < 256; ++bin) {
                if (!t[bin]) continue;
                if (M.v_of[bin] == 0xFFFFFFFFu) throw Error(CBLX_EDEVICE, "sharded build: a word fell into a bin no prefix maps to (internal error)");
                pcnt[d.piece * 256 + M.v_of[bin]] += t[bin];
                sum += t[bin];
            }
            if (sum != d.n) throw Error(CBLX_EDEVICE, "sharded build: the bin histogram counts " + std::to_string(sum) + " words, the slice has " + std::to_string(d.n) + " (internal error)");
        }
        deferred.clear();
        if (filled) {
            Records rec;
            rec.lo = std::move(a_lo);
            rec.hi = std::move(a_hi);
            rec.lo2 = Buf<u64>(c->pool, filled + 2);
            rec.hi2 = Buf<u8>(c->pool, OHS ? (filled + 2) * OHS : 8);
            PieceInput pin;
            pin.np = (u32)pbase.size();
            pin.cnt = pcnt.data();
            pin.pbase = pbase.data();
            pin.dig = &a_dig;
            pipeline<C>(c, rec, filled, Buf<u32>(), &pin);
            c->kmers_inserted += filled;
        }
        a_lo.reset(); a_hi.reset(); a_dig.reset();
        pcnt.clear(); pbase.clear();
        filled = 0;
    };
    for (u32 s = 0; s < nslices; ++s) {
        const u64 a = cuts[s], b = cuts[s + 1];
        before_slice(s);
        reap_sent();
        // -- KRN-1 with the bin histogram fused in, column prefixes of pass A
        ChunkPlan pl;
        BaseView pb = d_bases;
        u64 N = 0;
        if (b > a) { plan_chunks(c, pb, d_offsets + a, b - a, pl); N = pl.n_kmers; }  // (ends with a stream synchronisation)
        else CBLX_HIP(hipStreamSynchronize(c->stream));
        prev_work = Work();  // the previous slice's kernels are done: its workspace goes back to the pool
        if (N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "one slice takes fewer than 2^32-16 words (use more slices)");
        const u32 ntiles = (u32)ceil_div(N, RDX_TILE);
        Work wk;
        Buf<u64>& t_lo = wk.t_lo;
        Buf<u8>& t_hi = wk.t_hi;
        Buf<u32>&counts = wk.counts, &colpre = wk.colpre, &scratch = wk.scratch, &adj = wk.adj;
        Buf<u32> coltot(c->pool, 256);
        adj = Buf<u32>(c->pool, 256);
        wk.pl = std::move(pl);
        const ChunkPlan& plr = wk.pl;
        std::vector<u32> tot(256, 0u);
        if (N) {
            const size_t hs = hi_elem_size(P);
            t_lo = Buf<u64>(c->pool, N + 2);
            t_hi = Buf<u8>(c->pool, hs ? (N + 2) * hs : 8);
            counts = Buf<u32>(c->pool, (size_t)256 * (ntiles + 2));
            colpre = Buf<u32>(c->pool, (size_t)256 * ntiles);
            CBLX_HIP(hipMemsetAsync(counts.get(), 0, (size_t)256 * (ntiles + 2) * 4, c->stream));
            EncHist eh = eh0;
            eh.counts = counts.get();
            encode<C>(c, pb, plr, t_lo.get(), (HiT*)t_hi.get(), 0, eh);
            { StageTimer t(c, ST_SCAN);
              colscan(c, counts.get(), nullptr, ntiles, colpre.get(), coltot.get(), scratch);
              hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, colpre.get(), coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                 (const u32*)nullptr, ntiles, 1u, adj.get()); }
            CBLX_HIP(hipGetLastError());
            if (W > 1) tot = d2h_vec<u32>(c, coltot.get(), 256);
        }
        // -- what goes where: per destination the word count and the count of every segment (header of the exchange)
        const size_t HDR = 257;
        std::vector<u64> send((size_t)W * HDR, 0), recv((size_t)W * HDR, 0);
        if (W == 1) { send[0] = N; recv[0] = N; }  // (the segment counts follow at the end of the round: `deferred`)
        else
        for (u32 bin = 0; bin < 256; ++bin) {
            if (!tot[bin]) continue;
            if (M.v_of[bin] == 0xFFFFFFFFu) throw Error(CBLX_EDEVICE, "sharded build: a word fell into a bin no prefix maps to (internal error)");
            send[(size_t)M.d_of[bin] * HDR] += tot[bin];
            send[(size_t)M.d_of[bin] * HDR + 1 + M.v_of[bin]] += tot[bin];
        }
        {
            u64 sum = 0;
            for (u32 d = 0; d < W; ++d) sum += send[(size_t)d * HDR];
            if (sum != N) throw Error(CBLX_EDEVICE, "sharded build: the bin histogram counts " + std::to_string(sum) + " words, the slice has " + std::to_string(N) + " (internal error)");
        }
        if (W > 1) T.all_to_all_u64(send.data(), recv.data(), HDR);
        u64 own_a = 0, incoming = 0;
        for (u32 d = 0; d < me; ++d) own_a += send[(size_t)d * HDR];
        const u64 own = send[(size_t)me * HDR], own_b = own_a + own;
        if (recv[(size_t)me * HDR] != own) throw Error(CBLX_EDEVICE, "sharded build: the count exchange returned another own count (transport error)");
        for (u32 r = 0; r < W; ++r) incoming += recv[(size_t)r * HDR];
        if (incoming >= LIMIT) throw Error(CBLX_ERANGE, "one slice of the job sends this rank 2^32 words or more (use more slices)");
        if (trace) fprintf(stderr, "[cblx bins] rank %u slice %u: N=%llu own=%llu incoming=%llu filled=%llu cap=%llu mine=%llu job=%llu\n", me, s, (unsigned long long)N, (unsigned long long)own,
                           (unsigned long long)incoming, (unsigned long long)filled, (unsigned long long)cap, (unsigned long long)mine, (unsigned long long)job);
        if (filled + incoming > LIMIT) finish_round();
        if (!a_lo.get()) alloc_arena(cap, a_lo, a_hi, a_dig);
        if (filled + incoming > cap) grow(filled + incoming);
        // -- arena layout of the slice: the own piece first (pass A writes it there), then the other sources in rank order;
        //    the piece table keeps the logical order (slice-major, source-minor)
        std::vector<u64> so(W + 1, 0), ro(W + 1, 0);  // record positions in the send buffer / behind the own piece
        for (u32 d = 0; d < W; ++d) {
            so[d + 1] = so[d] + (d == me ? 0 : send[(size_t)d * HDR]);
            ro[d + 1] = ro[d] + (d == me ? 0 : recv[(size_t)d * HDR]);
        }
        for (u32 r = 0; r < W; ++r) {
            pbase.push_back((u32)(r == me ? filled : filled + own + ro[r]));
            for (u32 v = 0; v < 256; ++v) pcnt.push_back((u32)recv[(size_t)r * HDR + 1 + v]);
        }
        if (W == 1 && N) deferred.push_back(Deferred{pbase.size() - 1, N, std::move(coltot)});
        Sent S;
        const u64 nsend = N - own;
        S.lo = Buf<u64>(c->pool, nsend + 2);
        S.hi = Buf<u8>(c->pool, OHS ? (nsend + 2) * OHS : 8);
        S.dig = Buf<u8>(c->pool, nsend + 64);
        if (N) {
            const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
            const OwnWindow ow{own_a, own_b, a_lo.get() + filled, OHS ? (void*)(a_hi.get() + filled * OHS) : nullptr, a_dig.get() + filled};
            StageTimer t(c, ST_SCATTER);
            c->stages[ST_SCATTER].units += N;
            hipLaunchKernelGGL((k_radix_scatter<HiT, OutH, DigitBin, true>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, (const u64*)t_lo.get(), (const HiT*)t_hi.get(), tv, fn,
                               (const u32*)colpre.get(), (const u32*)adj.get(), S.lo.get(), (OutH*)S.hi.get(), nextd, S.dig.get(), (u32*)nullptr, 0u, 0u, 0u, (u32*)nullptr, 0u, ow);
            CBLX_HIP(hipGetLastError());
        }
        // -- the exchange (on the transport's stream, behind pass A): records, hi parts, digits
        u8* dst_lo = (u8*)(a_lo.get() + filled + own);
        std::vector<u64> sob(W + 1), rob(W + 1);
        auto xchg = [&](const u8* src, u8* dst, size_t es) {
            for (u32 d = 0; d <= W; ++d) { sob[d] = so[d] * es; rob[d] = ro[d] * es; }
            T.exchange(src, sob.data(), dst, rob.data(), c->stream);
        };
        if (W > 1) {
            xchg((const u8*)S.lo.get(), dst_lo, 8);
            if (OHS) xchg(S.hi.get(), a_hi.get() + (filled + own) * OHS, OHS);
            xchg(S.dig.get(), a_dig.get() + filled + own, 1);
        }
        filled += incoming;
        if (W > 1) {
            CBLX_HIP(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
            T.record(S.done, c->stream);
        }
        sent.push_back(std::move(S));
        // the slice's workspace (chunk plan, words, count matrix) stays until the next slice's chunk plan has synchronised the
        // stream: the host queues the next slice's first kernels while this slice's pass A is still running
        prev_work = std::move(wk);
    }
    finish_round();
    CBLX_HIP(hipStreamSynchronize(c->stream));
}

// ---- protocol "bins" with a GROUPED receiver: the wire hidden behind the receiver's own kernels ---------------------------------
// In sharded_insert_bins the remaining passes and the bucket kernels start when the LAST record has landed: at 8 GPUs the wire
// (25 - 30 ms) lies bare in front of 27 - 36 ms of receiver kernels. Here every rank's prefix range is cut into G groups of about
// equal sampled mass (choose_group_cuts) and the senders' first pass runs on bins that refine the pass-A segment by ALL cuts
// (DigitCut) — so a sender's output is contiguous per (destination, group), and inside that per segment, as before. The senders
// finish KRN-1 + pass A of all their slices first (10 - 12 ms during which the wire idles, but the GPU does not); then the data
// crosses GROUP-MAJOR: one grouped send / receive per group, carrying that group's share of every slice. The receive log is laid
// out exactly as before (slice-major, source-minor pieces, each sorted by bin), only the order of arrival changes: when group g
// has landed (an event on the transport's stream) its records — pieces of the log — go through the LSD passes, their window of the
// directory and the bucket kernels into the group's slot of the final arena (pipeline_group) while groups g + 1 .. are still on the
// wire. Per group the receiver's kernels take about as long as the group's bytes need on 7 links, so from the first group on the
// GPU is the bound. Inside a group a segment value occurs in one bin only (every cut is a group or rank boundary), so the group's
// piece table is the old one with other counts. Declines (returns false, identically on every rank: the decision rests on
// replicated values only) when the index is not empty anywhere, the cuts do not fit the table, or a rank's share needs two rounds.
__global__ void k_add_u64(u64* __restrict__ v, u64 n, u64 add) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += add;
}
// batches from this many k-mers (an upper bound of them) on take the FINE-bins build on one GPU; CBLX_FINE_MIN overrides (tests: 0)
u64 fine_min_words() {  // read per call: tests switch it
    const char* e = std::getenv("CBLX_FINE_MIN");
    return e ? std::strtoull(e, nullptr, 10) : (u64)(4u << 20);
}
// What CBLX_PROTO_AUTO resolves to on `world` ranks (cblx.h): between 2 and 4 GPUs every pair shares ONE link and the words of "bins" / "sorted" are
// bound by it — on 2 and 3 ranks the reads themselves cross ("replicate": 49.8 / 61.9 ms per step at cfg 3 and 55 GB/s per link against 101.3 / 76.7
// "sorted" and 127 / 89.8 "bins", profiles/r06_wire_emulated.md); on 4 the W encodes of "replicate" cost more than the packed suffixes of "sorted"
// (74.7 against 69.1); from 5 ranks on the receiver's kernels are the bound and "bins" adds no pass.
inline u32 auto_protocol(u32 world) { return world >= 2 && world <= 3 ? CBLX_PROTO_REPLICATE : (world == 4 ? CBLX_PROTO_SORTED : CBLX_PROTO_BINS); }
inline u32 recv_groups_wanted(const cblx_comm* cm) {
    if (cm->recv_groups) return cm->recv_groups;
    const char* e = std::getenv("CBLX_RECV_GROUPS");
    const u32 v = e ? (u32)std::strtoul(e, nullptr, 10) : 0;
    if (v) return std::min(v, 14u);
    // "replicate" has no wire for its groups to hide: two of them (the FINE bins still want the dense part of the range apart from the tail) — 48.2
    // against 49.0 ms with four at cfg 3 on two ranks
    const u32 proto = cm->protocol == CBLX_PROTO_AUTO && cm->t ? auto_protocol(cm->t->world) : cm->protocol;
    if (proto == CBLX_PROTO_REPLICATE) return 2u;
    return 4u;  // measured against a paced wire (profiles/r04_wire_emulated.md): 4 groups are best at 55 GB/s per link for every configuration; more pay on slower links
}
// The "replicate" protocol (sharded_insert_replicate below): the READS of every rank as this rank holds them after the all-gather — the own ones
// as the caller's ASCII bytes, the peers' as bit planes indexed by the sender's base positions — with every rank's slice cuts.
struct ReplicaSource {
    BaseView view{nullptr, nullptr, nullptr};
    const u64* d_off = nullptr;  // nseq + 1 offsets (the sender's own positions)
    u64 nseq = 0;
    std::vector<u64> cuts;       // nslices + 1 sequence indices relative to d_off
};
struct Replica {
    std::vector<ReplicaSource> src;
    // every slice in `parts` parts (the cuts of a source then hold nslices * parts + 1 entries): the peers' planes cross the links part by part and
    // a peer's part is transformed as soon as part_ready[t] has fired (null / empty: nothing to wait for). Stream order stays slice-major, rank-minor,
    // part-minor: parts only cut the pieces, they do not reorder them.
    u32 parts = 1;
    std::vector<hipEvent_t> part_ready;
    // a batch that is still ARRIVING (one rank, insert_device_sliced: slices land over PCIe): every slice is planned on its own once `ready(s)` has
    // made the stream wait for it (s = ~0u: the offsets' first and last entry)
    bool per_slice = false;
    std::function<void(u32)> ready;
};
// `single`: ONE rank whose "groups" are the part of the prefix space the FINE bins cover in blocks of 2^16 prefixes and the rest (insert_device_fine):
// no wire, the same sender and receiver steps
// `rep`: no records cross the wire — piece (slice, source) is what THIS rank's first pass keeps of the source's reads (its own prefix range), the
// receiver steps are the same
template <typename C>
bool sharded_insert_grouped(cblx_ctx* c, cblx_comm* cm, const u8* d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, const u32* bounds, bool single = false,
                            const Replica* rep = nullptr) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    constexpr bool DROP_HI = std::is_same<HiT, u8>::value;
    typedef typename std::conditional<DROP_HI, NoHi, HiT>::type OutH;
    constexpr size_t OHS = HiTraits<OutH>::has ? sizeof(u64) : 0;
    Transport& T = *cm->t;
    const Consts& P = c->P;
    const u32 W = T.world, me = T.rank, RB = P.PB - 8;
    const u32 G = recv_groups_wanted(cm);
    cm->groups_used = 0;
    cm->groups_fine = 0;
    if ((W < 2 && !single) || W > MAX_DEST || G < 2 || P.PB < 9 || nslices == 0) return false;
    for (u32 s = 0; s < nslices; ++s) if (cuts[s + 1] < cuts[s] || cuts[s + 1] > n) throw Error(CBLX_EINVAL, "slice cuts must be ascending and at most n");
    // -- the group cuts that go with these bounds (chosen once per set of bounds: a sampled histogram of the first slice, all-reduced)
    const std::vector<u32> bvec(bounds, bounds + (W - 1));
    if (!single && cm->g_bounds != bvec) {
        std::vector<u64> hist = sampled_prefix_hist<C>(c, T, d_bases, d_offsets + cuts[0], cuts[1] - cuts[0]);
        weigh_tail_for_fine_bins(hist, P.PB, W, G, std::min(SPLIT_HIST_BITS, P.PB));
        cm->g_cuts = choose_group_cuts(hist, bounds, W, G, P.PB);
        cm->g_bounds = bvec;
    }
    // PREFIX_BITS > 24: FINE bins (cuts.hpp) — the first pass also cuts the narrow groups into aligned blocks of 2^16 prefixes, so their
    // receiver sorts 16 bits in two LSD passes instead of 20 in three (CBLX_FINE_BINS=0: the bins of the top 8 prefix bits, as below 25)
    FinePlan FM;
    {
        const char* fe = std::getenv("CBLX_FINE_BINS");
        // every bin must imply the prefix bits a record does not carry behind the first pass: 65..72-bit words travel as their low 64 bits
        const u32 lmax = DROP_HI ? std::min(24u, 64u - P.SB) : 24u;
        if (P.PB > 24 && !(fe && fe[0] == '0')) FM = make_fine_plan(P.PB, lmax, bounds, W, cm->g_cuts, single);
    }
    const bool fine = FM.ok;
    if (single && !fine) return false;
    const CutPlan LM = fine ? CutPlan() : make_cut_plan(P.PB, bounds, W, cm->g_cuts);
    const CutPlan& M = fine ? static_cast<const CutPlan&>(FM) : LM;
    // -- the job: k-mers per rank (upper bound), whether any rank holds an index already
    const u64 n0 = cuts[0], n1 = cuts[nslices];
    u64 mine = 0;
    if (rep && rep->ready) rep->ready(~0u);
    if (n1 > n0) {
        const u64 first = d2h<u64>(c, d_offsets + n0), last = d2h<u64>(c, d_offsets + n1);
        if (last < first) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
        const u64 sub = (n1 - n0) * (u64)(P.K - 1);
        mine = last - first > sub ? last - first - sub : 0;
    }
    const u64 LIMIT = 0xFFFFFFF0ull - 2 * RDX_TILE;
    // (every term of the decision below is replicated: a rank that alone holds 2^32 k-mers says so through the sum)
    // The digit side channel of the receiver's first LSD pass (1 byte per word) stays OFF the wire by default (round 5): at 8 GPUs and 55 GB/s per
    // link the step is bound by the wire from the first group on (DESIGN_HISTORY.md §5.8), the byte is a ninth of it, and the receiver's first histogram
    // reads the records instead (8 bytes per word where it read 1: about the millisecond the senders' byte stores cost). CBLX_WIRE_DIGITS=1 sends
    // it as rounds 3 - 4 did; one rank (no wire) always keeps it.
    const char* wd_env = std::getenv("CBLX_WIRE_DIGITS");
    const bool wire_dig = single || rep != nullptr || (wd_env && wd_env[0] == '1');
    // What crosses the wire follows from switches every rank reads in its OWN environment (CBLX_FINE_BINS, CBLX_WIRE_DIGITS, CBLX_FINE_TAIL_WEIGHT,
    // CBLX_RECV_GROUPS): they are job-wide. A rank started with other values would wait for items its peers never send, or bin records by another
    // table — so the resolved choices ride in the all-reduce below (sum and sum of squares: equal on every rank iff W x sum(v^2) = sum(v)^2).
    const u64 choice = (fine ? 1ull : 0ull) | (wire_dig ? 2ull : 0ull) | ((u64)G << 2) | (std::min<u64>(fine_tail_weight_pct(), 0xFFFFull) << 8);
    u64 agree[5] = {mine, c->res.count != 0 ? 1ull : 0ull, mine >= LIMIT ? 1ull : 0ull, choice, choice * choice};
    T.all_reduce_sum_u64(agree, 5);
    if (agree[4] * W != agree[3] * agree[3])
        throw Error(CBLX_EINVAL, "sharded build: the ranks resolved CBLX_FINE_BINS / CBLX_WIRE_DIGITS / CBLX_FINE_TAIL_WEIGHT / CBLX_RECV_GROUPS differently (they are job-wide: set them in every rank's environment)");
    const u64 job = agree[0];
    if (!M.ok || agree[1] != 0 || agree[2] != 0 || job / W + job / (2 * W) + (1u << 20) >= LIMIT) return false;
    if (single && mine < fine_min_words()) return false;  // (a small batch: the groups' fixed costs outweigh the pass saved)
    const bool trace = std::getenv("CBLX_TRACE_SHARDED") != nullptr;
    const LsdPlan LP = fine ? lsd_plan_bits(FINE_LEVEL) : lsd_plan(P, false);  // (the receiver runs LSD passes only: pipeline.hpp; FINE bins: the first digit is the same for 16 and 24 sorted bits)
    const DigitBits nextd = wire_dig ? DigitBits{P.SB + LP.sh[0], LP.wid[0]} : DigitBits{0, 0};
    Buf<u32> d_tab(c->pool, DigitCut::LDS_WORDS);  // CutCell[CUT_KEYS], or u32[FINE_CELLS]: staged in LDS by the kernels that look bins up
    if (fine) h2d(c, d_tab.get(), FM.tab32.data(), FM.tab32.size());
    else h2d(c, d_tab.get(), reinterpret_cast<const u32*>(M.tab.data()), M.tab.size() * 2);
    DigitCut fn{P.SB, P.PB, RB, d_tab.get(), fine ? FM.ksh : 0xFFFFFFFFu};
    EncHist eh0{};
    eh0.nd = W; eh0.SB = P.SB; eh0.PB = P.PB; eh0.binRB = RB; eh0.cut_tab = d_tab.get(); eh0.cut_ksh = fn.ksh;
    if (single && fine) {
        // one rank: the plan is regular — multiples of 2^16 up to the one group cut, multiples of 2^lmax above — and the bin is arithmetic
        // (checked against the plan's table, cell by cell; the table route stays for anything else)
        const u32 x = cm->g_cuts[0], lm = DROP_HI ? std::min(24u, 64u - P.SB) : 24u;
        bool regular = (x & 0xFFFFu) == 0;
        for (u32 k = 0; regular && k < FINE_CELLS; ++k) {
            const u32 p = k << FM.ksh, b = ((p < x ? p : x) >> 16) + (p >= x ? (p >> lm) - (x >> lm) : 0u);
            regular = (FM.tab32[k] >> 8) == FINE_NO_CUT && (FM.tab32[k] & 255u) == b;
        }
        if (regular) { fn.reg_x = x; fn.reg_lmax = lm; fn.tab = nullptr; eh0.reg_x = x; eh0.reg_lmax = lm; eh0.cut_tab = nullptr; }
    }
    const u32 my_lo = M.bin_lo[me], my_cells = M.bin_lo[me + 1] - M.bin_lo[me], NG = M.ngroups[me];
    // cells (= my bins) of every one of my groups
    // (a cut that falls exactly on a segment boundary leaves one bin number unused: such a cell holds nothing and belongs nowhere)
    std::vector<u32> gc0(NG + 1, my_cells);
    for (u32 cl = my_cells; cl-- > 0;) if (M.iv_of[my_lo + cl] != 0xFFFFFFFFu) gc0[M.grp_of[M.iv_of[my_lo + cl]]] = cl;
    for (int g = (int)NG - 1; g >= 0; --g) if (gc0[g] > gc0[g + 1]) gc0[g] = gc0[g + 1];
    u32 Gmax = 0;
    for (u32 d = 0; d < W; ++d) Gmax = std::max(Gmax, M.ngroups[d]);

    // -- senders: KRN-1 + pass A of every slice; the own records go straight into the log, the rest waits in the slice's send buffer
    u64 cap = std::min<u64>(std::max<u64>(std::min<u64>(job, job / W + job / (4 * W) + (1u << 20)), 1024), LIMIT);
    Buf<u64> a_lo;
    Buf<u8> a_hi, a_dig;
    auto alloc_log = [&](u64 ncap, Buf<u64>& lo, Buf<u8>& hi, Buf<u8>& dg) {
        lo = Buf<u64>(c->pool, ncap + 2);
        hi = Buf<u8>(c->pool, OHS ? (ncap + 2) * OHS : 8);
        dg = Buf<u8>(c->pool, wire_dig ? ncap + 64 : 64);
    };
    struct Sent { Buf<u64> lo; Buf<u8> hi, dig; std::vector<u32> tot; u64 own_a = 0, own = 0; };
    std::vector<Sent> sent(nslices);
    std::vector<u32> pcnt, pbase;  // [piece][256] records per CELL of mine, log position of every piece; piece = slice * W + source
    const size_t HDR = 257;
    std::vector<std::vector<u64>> recvh(nslices);  // headers: what every source sends me, per slice
    u64 filled = 0;
    struct Drain { Transport& t; ~Drain() { try { t.wait(); } catch (...) {} } } drain{T};
    auto grow = [&](u64 need) {  // (only own pieces and the first group's early shares are in the log)
        T.wait();
        CBLX_HIP(hipStreamSynchronize(c->stream));
        const u64 ncap = std::min<u64>(LIMIT, need + need / 4 + 4096);
        Buf<u64> lo;
        Buf<u8> hi, dg;
        alloc_log(ncap, lo, hi, dg);
        if (filled) {
            CBLX_HIP(hipMemcpyAsync(lo.get(), a_lo.get(), filled * 8, hipMemcpyDeviceToDevice, c->stream));
            if (OHS) CBLX_HIP(hipMemcpyAsync(hi.get(), a_hi.get(), filled * OHS, hipMemcpyDeviceToDevice, c->stream));
            if (wire_dig) CBLX_HIP(hipMemcpyAsync(dg.get(), a_dig.get(), filled, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipStreamSynchronize(c->stream));
        }
        a_lo = std::move(lo); a_hi = std::move(hi); a_dig = std::move(dg);
        cap = ncap;
    };
    std::vector<hipEvent_t> gev(Gmax, nullptr);
    struct Events { std::vector<hipEvent_t>& v; ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } events{gev};
    // record range [first, first + len) of bins [b0, b1) in a slice's send buffer (the own window is not in it)
    auto send_range = [&](const Sent& S, u32 b0, u32 b1, u64& first, u64& len) {
        u64 before = 0, inside = 0;
        for (u32 bin = 0; bin < b1; ++bin) (bin < b0 ? before : inside) += S.tot[bin];
        first = before >= S.own_a + S.own ? before - S.own : before;  // ranges of other ranks lie wholly before or behind the own window
        len = inside;
    };
    // the items of (group k, slice s): this rank's records for every other rank's group k, and where the other ranks' records for
    // this rank's group k land in the log (every rank issues the same items in the same order, empty ones included: a callback
    // transport's exchange is a collective)
    auto add_items = [&](u32 k, u32 s, std::vector<Transport::Item>& items) {
        Sent& S = sent[s];
        std::vector<u64> so(W, 0), sl(W, 0), rof(W, 0), rl(W, 0);
        for (u32 d = 0; d < W; ++d) {
            if (d == me || k >= M.ngroups[d]) continue;
            u32 b0 = 256, b1 = 0;  // bins of (rank d, group k)
            for (u32 bin = M.bin_lo[d]; bin < M.bin_lo[d + 1]; ++bin)
                if (M.iv_of[bin] != 0xFFFFFFFFu && M.grp_of[M.iv_of[bin]] == k) { b0 = std::min(b0, bin); b1 = std::max(b1, bin + 1); }
            if (b0 >= b1) continue;
            send_range(S, b0, b1, so[d], sl[d]);
        }
        if (k < NG)
            for (u32 r = 0; r < W; ++r) {
                if (r == me) continue;
                const u32* pc = pcnt.data() + ((size_t)s * W + r) * 256;
                u64 before = 0, inside = 0;
                for (u32 cl = 0; cl < gc0[k + 1]; ++cl) (cl < gc0[k] ? before : inside) += pc[cl];
                rof[r] = (u64)pbase[(size_t)s * W + r] + before;
                rl[r] = inside;
            }
        auto item = [&](const u8* src, u8* dst, size_t es) {
            Transport::Item it{src, dst, so, sl, rof, rl};
            for (u32 r = 0; r < W; ++r) { it.s_off[r] *= es; it.s_len[r] *= es; it.r_off[r] *= es; it.r_len[r] *= es; }
            items.push_back(std::move(it));
        };
        item((const u8*)S.lo.get(), (u8*)a_lo.get(), 8);
        if (OHS) item(S.hi.get(), a_hi.get(), OHS);
        if (wire_dig) item(S.dig.get(), a_dig.get(), 1);
    };
    struct Work { ChunkPlan pl; Buf<u64> t_lo; Buf<u8> t_hi; Buf<u32> counts, colpre, scratch, adj, coltot; };
    Work prev_work;
    // ONE chunk plan for the call, the slices are ranges of it (a plan per slice cost the send phase 0.4 ms of host round trips each)
    ChunkPlan PL;
    BaseView pb = ascii_view(d_bases);
    std::vector<PlanSlice> psl(nslices);
    if (rep) {
        // -- "replicate": KRN-1 + the first pass over EVERY rank's reads, piece by piece; the pass keeps the records of this rank's prefix range (they
        //    go straight into the log) and drops the others. The own pieces first: the peers' planes are still crossing the links meanwhile.
        if (rep->src.size() != W) throw Error(CBLX_EINVAL, "replicate: one source per rank (internal error)");
        const u32 Q = std::max(1u, rep->parts), NT = nslices * Q;  // parts per slice, parts per source
        pcnt.assign((size_t)NT * W * 256, 0u);
        pbase.assign((size_t)NT * W, 0u);
        alloc_log(cap, a_lo, a_hi, a_dig);
        u64 over = 0;
        struct Late { size_t piece; u64 n; Buf<u32> coltot; };
        std::vector<Late> late;
        for (u32 r = 0; r < W; ++r) if (rep->src[r].cuts.size() != (size_t)NT + 1) throw Error(CBLX_EINVAL, "replicate: slice cuts of a source (internal error)");
        // the own source is planned once (its parts are ranges of the plan) and goes first: the peers' planes are crossing the links meanwhile;
        // then the peers part-major — exchange t carries part t of every peer — each part planned on its own once it has landed
        ChunkPlan PLme;
        std::vector<PlanSlice> pslme(NT);
        BaseView vme = rep->src[me].view;
        const bool own_whole = !rep->per_slice && rep->src[me].nseq != 0;
        if (own_whole) plan_chunks(c, vme, rep->src[me].d_off, rep->src[me].nseq, PLme, nullptr, &rep->src[me].cuts, &pslme);
        std::vector<std::pair<u32, u32>> order;  // (source, part)
        for (u32 t = 0; t < NT; ++t) order.push_back({me, t});
        for (u32 t = 0; t < NT; ++t) for (u32 r = 0; r < W; ++r) if (r != me) order.push_back({r, t});
        {
            for (const auto& rt : order) {
                if (over) break;
                const u32 r = rt.first, t = rt.second, s = t / Q;
                const ReplicaSource& R = rep->src[r];
                if (R.nseq == 0) continue;
                prev_work = Work();
                Work wk;
                const ChunkPlan* plan = &PLme;
                const PlanSlice* part = &pslme[t];
                BaseView vs = vme;
                if (!(r == me && own_whole)) {  // the part is planned once it has landed
                    if (rep->ready) rep->ready(t);
                    if (r != me) {
                        if (t < rep->part_ready.size() && rep->part_ready[t]) CBLX_HIP(hipStreamWaitEvent(c->stream, rep->part_ready[t], 0));
                        else T.wait();
                    }
                    if (R.cuts[t + 1] == R.cuts[t]) continue;
                    vs = R.view;
                    plan_chunks(c, vs, R.d_off + R.cuts[t], R.cuts[t + 1] - R.cuts[t], wk.pl);
                    plan = &wk.pl;
                    part = nullptr;
                }
                const u64 N = part ? part->k_hi - part->k_lo : plan->n_kmers;
                if (N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "one slice takes fewer than 2^32-16 words (use more slices)");
                if (!N) continue;
                const u32 ntiles = (u32)ceil_div(N, RDX_TILE);
                wk.coltot = Buf<u32>(c->pool, 256);
                wk.adj = Buf<u32>(c->pool, 256);
                const size_t hs = hi_elem_size(P);
                wk.t_lo = Buf<u64>(c->pool, N + 2);
                wk.t_hi = Buf<u8>(c->pool, hs ? (N + 2) * hs : 8);
                wk.counts = Buf<u32>(c->pool, (size_t)256 * (ntiles + 2));
                wk.colpre = Buf<u32>(c->pool, (size_t)256 * ntiles);
                CBLX_HIP(hipMemsetAsync(wk.counts.get(), 0, (size_t)256 * (ntiles + 2) * 4, c->stream));
                EncHist eh = eh0;
                eh.counts = wk.counts.get();
                encode<C>(c, vs, *plan, wk.t_lo.get(), (HiT*)wk.t_hi.get(), 0, eh, part);
                { StageTimer t(c, ST_SCAN);
                  colscan(c, wk.counts.get(), nullptr, ntiles, wk.colpre.get(), wk.coltot.get(), wk.scratch);
                  hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, wk.colpre.get(), wk.coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                     (const u32*)nullptr, ntiles, 1u, wk.adj.get()); }
                CBLX_HIP(hipGetLastError());
                const size_t piece = ((size_t)s * W + r) * Q + (t % Q);
                if (single) {
                    // one rank: every record is its own and the pass writes the log directly; the bin counts of the piece stay on the device until
                    // the last slice is through (no host round trip between KRN-1 and the first pass: the next slice may be landing)
                    if (filled + N >= LIMIT) { over = 1; break; }
                    if (filled + N > cap) grow(filled + N);
                    pbase[piece] = (u32)filled;
                    const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
                    { StageTimer t(c, ST_SCATTER);
                      c->stages[ST_SCATTER].units += N;
                      hipLaunchKernelGGL((k_radix_scatter<HiT, OutH, DigitCut, false>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, (const u64*)wk.t_lo.get(), (const HiT*)wk.t_hi.get(), tv, fn,
                                         (const u32*)wk.colpre.get(), (const u32*)wk.adj.get(), a_lo.get() + filled, OHS ? (OutH*)(a_hi.get() + filled * OHS) : (OutH*)nullptr, nextd, a_dig.get() + filled); }
                    CBLX_HIP(hipGetLastError());
                    late.push_back(Late{piece, N, std::move(wk.coltot)});
                    filled += N;
                    prev_work = std::move(wk);
                    continue;
                }
                const std::vector<u32> tot = d2h_vec<u32>(c, wk.coltot.get(), 256);
                u64 sum = 0, own_a = 0, own = 0;
                for (u32 bin = 0; bin < 256; ++bin) {
                    if (!tot[bin]) continue;
                    if (M.iv_of[bin] == 0xFFFFFFFFu) throw Error(CBLX_EDEVICE, "sharded build: a word fell into a bin no prefix maps to (internal error)");
                    const u32 d = M.dest_of[M.iv_of[bin]];
                    if (d < me) own_a += tot[bin];
                    if (d == me) { own += tot[bin]; pcnt[piece * 256 + (bin - my_lo)] = tot[bin]; }
                    sum += tot[bin];
                }
                if (sum != N) throw Error(CBLX_EDEVICE, "sharded build: the bin histogram counts " + std::to_string(sum) + " words, the piece has " + std::to_string(N) + " (internal error)");
                if (filled + own >= LIMIT) { over = 1; break; }  // (settled with the other ranks below: the job leaves together)
                if (filled + own > cap) grow(filled + own);
                pbase[piece] = (u32)filled;
                {
                    const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
                    const OwnWindow ow{own_a, own_a + own, a_lo.get() + filled, OHS ? (void*)(a_hi.get() + filled * OHS) : nullptr, a_dig.get() + filled};
                    StageTimer t(c, ST_SCATTER);
                    c->stages[ST_SCATTER].units += N;
                    hipLaunchKernelGGL((k_radix_scatter<HiT, OutH, DigitCut, true>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, (const u64*)wk.t_lo.get(), (const HiT*)wk.t_hi.get(), tv, fn,
                                       (const u32*)wk.colpre.get(), (const u32*)wk.adj.get(), (u64*)nullptr, (OutH*)nullptr, nextd, (u8*)nullptr, (u32*)nullptr, 0u, 0u, 0u, (u32*)nullptr, 0u, ow);
                    CBLX_HIP(hipGetLastError());
                }
                if (trace) fprintf(stderr, "[cblx replicate] rank %u piece (slice %u, source %u, part %u): N=%llu kept=%llu filled=%llu\n", me, s, r, t % Q, (unsigned long long)N,
                                   (unsigned long long)own, (unsigned long long)filled);
                filled += own;
                prev_work = std::move(wk);
            }
            CBLX_HIP(hipStreamSynchronize(c->stream));  // the own source's plan dies here
        }
        T.wait();
        T.all_reduce_sum_u64(&over, 1);
        if (over) throw Error(CBLX_ERANGE, "a rank's share of the job takes more than one round: use more ranks");
        for (Late& l : late) {  // one rank: the bin counts of every piece, read once the passes are queued
            const std::vector<u32> t = d2h_vec<u32>(c, l.coltot.get(), 256);
            u64 sum = 0;
            for (u32 bin = 0; bin < 256; ++bin) {
                if (!t[bin]) continue;
                if (M.iv_of[bin] == 0xFFFFFFFFu || bin < my_lo || bin - my_lo >= my_cells) throw Error(CBLX_EDEVICE, "sharded build: a word fell into a bin no prefix maps to (internal error)");
                pcnt[l.piece * 256 + (bin - my_lo)] = t[bin];
                sum += t[bin];
            }
            if (sum != l.n) throw Error(CBLX_EDEVICE, "sharded build: the bin histogram counts " + std::to_string(sum) + " words, the piece has " + std::to_string(l.n) + " (internal error)");
        }
    } else {
    if (n1 > n0) {
        std::vector<u64> marks(nslices + 1);
        for (u32 s = 0; s <= nslices; ++s) marks[s] = cuts[s] - n0;
        plan_chunks(c, pb, d_offsets + n0, n1 - n0, PL, nullptr, &marks, &psl);
    }
    for (u32 s = 0; s < nslices; ++s) {
        const u64 N = psl[s].k_hi - psl[s].k_lo;
        prev_work = Work();  // (its kernels are queued on this stream in front of whatever takes the blocks next)
        if (N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "one slice takes fewer than 2^32-16 words (use more slices)");
        const u32 ntiles = (u32)ceil_div(N, RDX_TILE);
        Work wk;
        wk.coltot = Buf<u32>(c->pool, 256);
        wk.adj = Buf<u32>(c->pool, 256);
        Sent& S = sent[s];
        S.tot.assign(256, 0u);
        if (N) {
            const size_t hs = hi_elem_size(P);
            wk.t_lo = Buf<u64>(c->pool, N + 2);
            wk.t_hi = Buf<u8>(c->pool, hs ? (N + 2) * hs : 8);
            wk.counts = Buf<u32>(c->pool, (size_t)256 * (ntiles + 2));
            wk.colpre = Buf<u32>(c->pool, (size_t)256 * ntiles);
            CBLX_HIP(hipMemsetAsync(wk.counts.get(), 0, (size_t)256 * (ntiles + 2) * 4, c->stream));
            EncHist eh = eh0;
            eh.counts = wk.counts.get();
            encode<C>(c, pb, PL, wk.t_lo.get(), (HiT*)wk.t_hi.get(), 0, eh, &psl[s]);
            { StageTimer t(c, ST_SCAN);
              colscan(c, wk.counts.get(), nullptr, ntiles, wk.colpre.get(), wk.coltot.get(), wk.scratch);
              hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, wk.colpre.get(), wk.coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                 (const u32*)nullptr, ntiles, 1u, wk.adj.get()); }
            CBLX_HIP(hipGetLastError());
            S.tot = d2h_vec<u32>(c, wk.coltot.get(), 256);
        }
        // header per destination: words, then words per cell of the destination's range
        std::vector<u64> send((size_t)W * HDR, 0);
        recvh[s].assign((size_t)W * HDR, 0);
        u64 sum = 0;
        for (u32 bin = 0; bin < 256; ++bin) {
            if (!S.tot[bin]) continue;
            if (M.iv_of[bin] == 0xFFFFFFFFu) throw Error(CBLX_EDEVICE, "sharded build: a word fell into a bin no prefix maps to (internal error)");
            const u32 d = M.dest_of[M.iv_of[bin]];
            send[(size_t)d * HDR] += S.tot[bin];
            send[(size_t)d * HDR + 1 + (bin - M.bin_lo[d])] += S.tot[bin];
            sum += S.tot[bin];
        }
        if (sum != N) throw Error(CBLX_EDEVICE, "sharded build: the bin histogram counts " + std::to_string(sum) + " words, the slice has " + std::to_string(N) + " (internal error)");
        T.all_to_all_u64(send.data(), recvh[s].data(), HDR);
        const std::vector<u64>& recv = recvh[s];
        u64 incoming = 0;
        for (u32 d = 0; d < me; ++d) S.own_a += send[(size_t)d * HDR];
        S.own = send[(size_t)me * HDR];
        if (recv[(size_t)me * HDR] != S.own) throw Error(CBLX_EDEVICE, "sharded build: the count exchange returned another own count (transport error)");
        for (u32 r = 0; r < W; ++r) incoming += recv[(size_t)r * HDR];
        {   // a share that needs two rounds is an error of the JOB: every rank learns of it here and leaves together (a rank that threw on
            // its own left its peers waiting in the next collective until the launcher's deadline)
            u64 over = filled + incoming >= LIMIT ? 1ull : 0ull;
            T.all_reduce_sum_u64(&over, 1);
            if (over) throw Error(CBLX_ERANGE, "a rank's share of the job takes more than one round: set CBLX_RECV_GROUPS=1 (ungrouped receiver) or use more ranks");
        }
        if (!a_lo.get()) alloc_log(cap, a_lo, a_hi, a_dig);
        if (filled + incoming > cap) grow(filled + incoming);
        // log layout of the slice as in sharded_insert_bins: the own piece first, then the other sources in rank order
        u64 ro = 0;
        for (u32 r = 0; r < W; ++r) {
            pbase.push_back((u32)(r == me ? filled : filled + S.own + ro));
            if (r != me) ro += recv[(size_t)r * HDR];
            for (u32 cl = 0; cl < 256; ++cl) pcnt.push_back((u32)recv[(size_t)r * HDR + 1 + cl]);
        }
        const u64 nsend = N - S.own;
        S.lo = Buf<u64>(c->pool, nsend + 2);
        S.hi = Buf<u8>(c->pool, OHS ? (nsend + 2) * OHS : 8);
        S.dig = Buf<u8>(c->pool, wire_dig ? nsend + 64 : 64);
        if (N) {
            const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
            const OwnWindow ow{S.own_a, S.own_a + S.own, a_lo.get() + filled, OHS ? (void*)(a_hi.get() + filled * OHS) : nullptr, wire_dig ? a_dig.get() + filled : (u8*)nullptr};
            StageTimer t(c, ST_SCATTER);
            c->stages[ST_SCATTER].units += N;
            if (single)  // one rank: every record is its own, the pass writes the log directly (measured: the redirecting instantiation costs the same)
                hipLaunchKernelGGL((k_radix_scatter<HiT, OutH, DigitCut, false>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, (const u64*)wk.t_lo.get(), (const HiT*)wk.t_hi.get(), tv, fn,
                                   (const u32*)wk.colpre.get(), (const u32*)wk.adj.get(), a_lo.get() + filled, OHS ? (OutH*)(a_hi.get() + filled * OHS) : (OutH*)nullptr, nextd, a_dig.get() + filled);
            else
            hipLaunchKernelGGL((k_radix_scatter<HiT, OutH, DigitCut, true>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, (const u64*)wk.t_lo.get(), (const HiT*)wk.t_hi.get(), tv, fn,
                               (const u32*)wk.colpre.get(), (const u32*)wk.adj.get(), S.lo.get(), (OutH*)S.hi.get(), nextd, wire_dig ? S.dig.get() : (u8*)nullptr, (u32*)nullptr, 0u, 0u, 0u, (u32*)nullptr, 0u, ow);
            CBLX_HIP(hipGetLastError());
        }
        if (trace) fprintf(stderr, "[cblx grouped] rank %u slice %u: N=%llu own=%llu incoming=%llu filled=%llu\n", me, s, (unsigned long long)N, (unsigned long long)S.own,
                           (unsigned long long)incoming, (unsigned long long)filled);
        filled += incoming;
        if (s + 1 < nslices) {  // the first group's share of this slice crosses the links under the next slice's kernels
            std::vector<Transport::Item> items;
            add_items(0, s, items);
            T.exchange_items(items, c->stream);
        }
        prev_work = std::move(wk);
    }
    // -- the wire, group-major: exchange k carries group k of EVERY rank (ranks with fewer groups send nothing in the later ones)
    for (u32 k = 0; k < Gmax; ++k) {
        std::vector<Transport::Item> items;
        // (group 0 of every slice but the last left right behind that slice's pass A, under the next slice's kernels)
        for (u32 s = (k == 0 ? nslices - 1 : 0u); s < nslices; ++s) add_items(k, s, items);
        T.exchange_items(items, c->stream);
        CBLX_HIP(hipEventCreateWithFlags(&gev[k], hipEventDisableTiming));
        T.record(gev[k], c->stream);
    }
    }  // (records on the wire)
    prev_work = Work();  // (the stream is past the last slice's pass A once the first group is waited for; the workspace is only returned to the pool)
    // -- the receiver, group by group behind the wire
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    Resident fin;
    fin.bv = Buf<u64>(c->pool, nwords);
    CBLX_HIP(hipMemsetAsync(fin.bv.get(), 0, nwords * 8, c->stream));
    fin.a_lo = Buf<u64>(c->pool, filled + 2);
    Buf<u64> fin_hi_keep;  // 16-byte records through the passes: a target for the hi parts even when the arena keeps none
    if (WS) fin.a_hi = Buf<u64>(c->pool, filled + 2);
    else if (OHS) fin_hi_keep = Buf<u64>(c->pool, filled + 2);
    u64* fin_hi = WS ? fin.a_hi.get() : fin_hi_keep.get();
    // words of every group of mine
    std::vector<u64> gN(NG, 0);
    const size_t np = pbase.size();  // nslices * W pieces (times the parts of a slice under "replicate")
    for (size_t p = 0; p < np; ++p)
        for (u32 g = 0; g < NG; ++g)
            for (u32 cl = gc0[g]; cl < gc0[g + 1]; ++cl) gN[g] += pcnt[p * 256 + cl];
    u64 gmax = 0;
    for (u64 x : gN) gmax = std::max(gmax, x);
    Buf<u64> scr_lo(c->pool, gmax + 2), scr_hi(c->pool, OHS ? gmax + 2 : 1);
    Buf<u8> dig2(c->pool, gmax + 64);
    std::vector<Resident> parts(NG);
    std::vector<u64> gbase(NG + 1, 0);
    // prefix window of every group: its cuts (multiples of 64), the range's ends rounded outwards
    auto group_first_prefix = [&](u32 g) -> u64 {
        if (g >= NG) return me + 1 < W ? ((u64)bounds[me] + 63) & ~63ull : nprefix;
        if (g == 0) return me ? (u64)bounds[me - 1] & ~63ull : 0ull;
        u32 cl = gc0[g];
        while (cl + 1 < my_cells && M.iv_of[my_lo + cl] == 0xFFFFFFFFu) ++cl;  // (an unused bin number in front of the group's first cell)
        const u32 iv = M.iv_of[my_lo + cl];
        return iv && iv != 0xFFFFFFFFu ? (u64)M.cuts[iv - 1] : 0ull;
    };
    for (u32 g = 0; g < NG; ++g) {
        gbase[g + 1] = gbase[g] + gN[g];
        if (g < Gmax && gev[g]) CBLX_HIP(hipStreamWaitEvent(c->stream, gev[g], 0));
        if (gN[g] == 0) continue;
        // the group's share of every piece: counts per SEGMENT (inside a group a segment occurs in one cell only), first record
        std::vector<u32> cnt_g(np * 256, 0u), pb_g(np), segp(256, 0xFFFFFFFFu);
        if (fine)  // a bin's records share their prefix bits from the bin's level up: the first prefix of its aligned block
            for (u32 cl = gc0[g]; cl < gc0[g + 1]; ++cl) {
                const u32 bin = my_lo + cl, iv = M.iv_of[bin];
                if (iv == 0xFFFFFFFFu) continue;
                // (the all-ones bin: the block of 2^sort_bits prefixes that ends at 2^PREFIX_BITS — its records OR their low bits back in and land
                // on the all-ones prefix; with the all-ones prefix ITSELF as the base every row of k_dir_gather's table for this segment
                // mapped to that one prefix, one of them with the real start and the others with EMPTY, unordered)
                const u32 sbg = FM.sort_bits[me][g];
                segp[FM.seg_of[bin]] = bin == 255 ? (u32)((((1ull << P.PB) - 1) >> sbg) << sbg) : (FM.first[iv] >> FM.level[iv]) << FM.level[iv];
            }
        for (size_t p = 0; p < np; ++p) {
            u64 before = 0;
            for (u32 cl = 0; cl < gc0[g]; ++cl) before += pcnt[p * 256 + cl];
            pb_g[p] = (u32)(pbase[p] + before);
            for (u32 cl = gc0[g]; cl < gc0[g + 1]; ++cl) {
                const u32 v = fine ? FM.seg_of[my_lo + cl] : M.v_of[my_lo + cl];
                if (v == 0xFFFFFFFFu) { if (pcnt[p * 256 + cl]) throw Error(CBLX_EDEVICE, "grouped receiver: words in a bin no prefix maps to (internal error)"); continue; }
                if (cnt_g[p * 256 + v] && pcnt[p * 256 + cl]) throw Error(CBLX_EDEVICE, "grouped receiver: a segment occurs in two cells of one group (internal error)");
                cnt_g[p * 256 + v] += pcnt[p * 256 + cl];
            }
        }
        PieceInput pin;
        pin.np = (u32)np;
        pin.cnt = cnt_g.data();
        pin.pbase = pb_g.data();
        if (wire_dig) { pin.dig_in = a_dig.get(); pin.dig_out = dig2.get(); }  // (else: the first histogram reads the records, the later passes' digits go to a buffer of the pass's own)
        if (fine) { pin.sort_bits = FM.sort_bits[me][g]; pin.seg_prefix = segp.data(); }
        DirWindow win;
        win.w_lo = (u32)group_first_prefix(g);
        u64 whi = group_first_prefix(g + 1);
        if (fine && whi > (nprefix >> 1) && win.w_lo < (nprefix >> 1)) {
            // FINE bins: no necklace prefix lies in the upper half of the prefix space except the all-ones word's (bin 255): a group that
            // reaches up there and holds no such word ends its directory window at the half (the dense directory costs a pass over the window)
            u64 ones = 0;
            if (my_lo + my_cells == 256) for (size_t p = 0; p < np; ++p) ones += pcnt[p * 256 + (255 - my_lo)];
            if (!ones) whi = nprefix >> 1;
        }
        win.w_hi = (u32)std::min<u64>(whi, 0xFFFFFFC0ull);
        win.bv = fin.bv.get();
        GroupRegions R;
        R.log_lo = a_lo.get();
        R.log_hi = OHS ? (const void*)a_hi.get() : nullptr;
        R.fin_lo = fin.a_lo.get() + gbase[g];
        R.fin_hi = fin_hi ? fin_hi + gbase[g] : nullptr;
        R.scr_lo = scr_lo.get();
        R.scr_hi = OHS ? scr_hi.get() : nullptr;
        if (trace) fprintf(stderr, "[cblx grouped] rank %u group %u: %llu words, prefixes [%u, %u), cells [%u, %u), %u bits sorted behind the first pass\n", me, g, (unsigned long long)gN[g], win.w_lo, win.w_hi,
                           gc0[g], gc0[g + 1], pin.sort_bits ? pin.sort_bits : RB);
        pipeline_group<C>(c, R, pin, gN[g], win, parts[g]);
        ++cm->groups_used;
        if (pin.sort_bits == FINE_LEVEL) ++cm->groups_fine;
    }
    T.wait();
    CBLX_HIP(hipStreamSynchronize(c->stream));
    for (Sent& S : sent) { S.lo.reset(); S.hi.reset(); S.dig.reset(); }
    a_lo.reset(); a_hi.reset(); a_dig.reset();
    // -- one index out of the groups: tables concatenated (starts made absolute), rank directory over the whole bitvector
    if (filled) {
        u64 nb = 0;
        for (const Resident& r : parts) { nb += r.nb; fin.count += r.count; }
        fin.nb = nb;
        fin.prefix = Buf<u32>(c->pool, nb + 1);
        fin.start = Buf<u64>(c->pool, nb + 1);
        fin.cnt = Buf<u32>(c->pool, nb + 1);
        fin.kind = Buf<u8>(c->pool, nb + 1);
        u64 at = 0;
        for (u32 g = 0; g < NG; ++g) {
            const Resident& r = parts[g];
            if (!r.nb) continue;
            CBLX_HIP(hipMemcpyAsync(fin.prefix.get() + at, r.prefix.get(), r.nb * 4, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(fin.start.get() + at, r.start.get(), r.nb * 8, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(fin.cnt.get() + at, r.cnt.get(), r.nb * 4, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(fin.kind.get() + at, r.kind.get(), r.nb, hipMemcpyDeviceToDevice, c->stream));
            if (gbase[g]) hipLaunchKernelGGL(k_add_u64, grid1(r.nb, 256), dim3(256), 0, c->stream, fin.start.get() + at, r.nb, gbase[g]);
            at += r.nb;
        }
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, fin.start.get() + nb, filled);
        Buf<u32> popc(c->pool, nwords);
        fin.rank_dir = Buf<u64>(c->pool, nwords + 1);
        hipLaunchKernelGGL(k_bv_or, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, (const u64*)fin.bv.get(), (const u64*)fin.bv.get(), fin.bv.get(), popc.get());
        const u64 nb2 = exclusive_scan<u64>(c, popc.get(), nwords, fin.rank_dir.get());
        CBLX_HIP(hipGetLastError());
        if (nb2 != nb) throw Error(CBLX_EDEVICE, "grouped receiver: the groups hold " + std::to_string(nb) + " buckets, the bitvector " + std::to_string(nb2) + " (internal error)");
        CBLX_HIP(hipStreamSynchronize(c->stream));
        parts.clear();
        c->res = std::move(fin);
        c->kmers_inserted += filled;
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    return true;
}

// ---- protocol "replicate" (round 6): ship READS, not words -----------------------------------------------------------------------------
// Between 2 - 4 GPUs every pair shares ONE xGMI link, and the words of the other protocols put 5 - 8 bytes per k-mer on it: at cfg 3 and 55 GB/s
// per link two ranks take 101 ms per step ("sorted"; 127 "bins") where ONE GPU builds the same reads in 39 (profiles/r05_wire_emulated.md). The
// k-mer -> word transform is a pure function of the bases, and the bases are 0.3 bytes per k-mer as bit planes (3 bits per base,
// 150 / 120 bases per k-mer): every rank packs ITS reads into planes on the device (k_pack_planes), the planes and the read offsets are
// all-gathered, and every rank runs KRN-1 and the first partition pass over ALL ranks' reads, keeping the words of its own prefix range
// (k_radix_scatter drops the others where it would have sent them) — W encodes per rank instead of one, nothing else added, and the job's stream
// order (slice-major, rank-minor) holds by construction: the pieces are the same (slice, source) pieces the "bins" receiver sees, made locally.
// The receiver steps (groups, FINE bins, directory windows, bucket kernels) are sharded_insert_grouped's own.
template <typename C>
bool sharded_insert_replicate(cblx_ctx* c, cblx_comm* cm, const u8* d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, const u32* bounds) {
    Transport& T = *cm->t;
    const u32 W = T.world, me = T.rank;
    if (W < 2 || nslices == 0) return false;
    check_aligned16(d_bases, "d_bases");
    for (u32 s = 0; s < nslices; ++s) if (cuts[s + 1] < cuts[s] || cuts[s + 1] > n) throw Error(CBLX_EINVAL, "slice cuts must be ascending and at most n");
    const u64 n0 = cuts[0], n1 = cuts[nslices], nseq = n1 - n0;
    // The planes cross the links in PARTS (two per call at least: a one-slice call is cut in halves of its reads) so that a peer's first part can be
    // transformed while its later ones are still on the wire; parts cut the (slice, source) pieces, they do not reorder them. (Every part of every
    // peer is planned on its own — 0.4 ms of host round trips each: four parts were measured too and cost what they hid at 55 GB/s per link.)
    const u32 Q = std::max(1u, 2u / nslices), NT = nslices * Q;
    std::vector<u64> xcuts(NT + 1), ox(NT + 1, 0);  // sequence index (relative to n0) and base offset at every part boundary
    for (u32 s = 0; s < nslices; ++s)
        for (u32 q = 0; q < Q; ++q) xcuts[s * Q + q] = cuts[s] - n0 + (cuts[s + 1] - cuts[s]) * q / Q;
    xcuts[NT] = nseq;
    if (nseq) {
        for (u32 t = 0; t <= NT; ++t) ox[t] = (t && xcuts[t] == xcuts[t - 1]) ? ox[t - 1] : d2h<u64>(c, d_offsets + n0 + xcuts[t]);
        for (u32 t = 0; t < NT; ++t) if (ox[t + 1] < ox[t]) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
    }
    const u64 first = ox[0], last = ox[NT];
    const u64 g0 = first >> 4, g1 = nseq ? (last + 15) >> 4 : g0, ng = g1 - g0;
    // -- what every rank holds: sequences, part boundaries (sequence index and base offset)
    const size_t per = 1 + 2 * ((size_t)NT + 1);
    std::vector<u64> send(per * W), recv(per * W);
    for (u32 d = 0; d < W; ++d) {
        u64* h = send.data() + d * per;
        h[0] = nseq;
        for (u32 t = 0; t <= NT; ++t) { h[1 + t] = xcuts[t]; h[2 + NT + t] = ox[t]; }
    }
    T.all_to_all_u64(send.data(), recv.data(), per);
    auto hdr_cut = [&](u32 r, u32 t) { return recv[r * per + 1 + t]; };
    auto hdr_off = [&](u32 r, u32 t) { return recv[r * per + 2 + NT + t]; };
    auto ga_of = [&](u32 r, u32 t) { return hdr_off(r, t) >> 4; };                                           // first plane group of part t of rank r
    auto gb_of = [&](u32 r, u32 t) { return recv[r * per] ? (hdr_off(r, t + 1) + 15) >> 4 : ga_of(r, t); };  // one past its last
    // -- own planes, then the all-gather of planes and offsets, part by part (the own pieces are transformed under it)
    Buf<u32> my_codes(c->pool, ng + 4);
    Buf<u16> my_valid(c->pool, ng + 4);
    if (ng) hipLaunchKernelGGL(k_pack_planes, grid1(ng, 256), dim3(256), 0, c->stream, d_bases, g0, g1, last, my_codes.get(), my_valid.get());
    CBLX_HIP(hipGetLastError());
    std::vector<u64> gro(W + 1, 0), sqo(W + 1, 0);  // plane groups / offset words in front of every peer's share of the gather buffers
    for (u32 r = 0; r < W; ++r) {
        for (u32 t = 0; t < NT; ++t) if (hdr_cut(r, t + 1) < hdr_cut(r, t) || hdr_off(r, t + 1) < hdr_off(r, t)) throw Error(CBLX_EDEVICE, "replicate: a rank announced part boundaries out of order (transport error)");
        if (hdr_cut(r, NT) != recv[r * per]) throw Error(CBLX_EDEVICE, "replicate: a rank's parts do not add up to its reads (transport error)");
        const u64 ngr = recv[r * per] ? gb_of(r, NT - 1) - ga_of(r, 0) : 0;
        gro[r + 1] = gro[r] + (r == me ? 0 : ngr + 4);  // (+ slack: the tile loads of KRN-1 read a few words ahead)
        sqo[r + 1] = sqo[r] + (r == me ? 0 : recv[r * per] + 1);
    }
    Buf<u32> all_codes(c->pool, gro[W] + 4);
    Buf<u16> all_valid(c->pool, gro[W] + 4);
    Buf<u64> all_off(c->pool, sqo[W] + 2);
    CBLX_HIP(hipMemsetAsync(all_codes.get(), 0, (gro[W] + 4) * 4, c->stream));
    CBLX_HIP(hipMemsetAsync(all_valid.get(), 0, (gro[W] + 4) * 2, c->stream));
    std::vector<hipEvent_t> ev(NT, nullptr);
    struct Events { std::vector<hipEvent_t>& v; ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } events{ev};
    struct Drain { Transport& t; ~Drain() { try { t.wait(); } catch (...) {} } } drain{T};
    for (u32 t = 0; t < NT; ++t) {
        auto planes = [&](const void* src, void* dst, size_t es) {
            Transport::Item it{(const u8*)src, (u8*)dst, std::vector<u64>(W, 0), std::vector<u64>(W, 0), std::vector<u64>(W, 0), std::vector<u64>(W, 0)};
            for (u32 r = 0; r < W; ++r) {
                if (r == me) continue;
                it.s_off[r] = (ga_of(me, t) - ga_of(me, 0)) * es;
                it.s_len[r] = (gb_of(me, t) - ga_of(me, t)) * es;
                it.r_off[r] = (gro[r] + ga_of(r, t) - ga_of(r, 0)) * es;
                it.r_len[r] = (gb_of(r, t) - ga_of(r, t)) * es;
            }
            return it;
        };
        std::vector<Transport::Item> items;
        items.push_back(planes(my_codes.get(), all_codes.get(), 4));
        items.push_back(planes(my_valid.get(), all_valid.get(), 2));
        if (t == 0) {  // every read offset rides with the first part
            Transport::Item it{(const u8*)(d_offsets + n0), (u8*)all_off.get(), std::vector<u64>(W, 0), std::vector<u64>(W, 0), std::vector<u64>(W, 0), std::vector<u64>(W, 0)};
            for (u32 r = 0; r < W; ++r) {
                if (r == me) continue;
                it.s_len[r] = nseq ? (nseq + 1) * 8 : 0;
                it.r_off[r] = sqo[r] * 8;
                it.r_len[r] = recv[r * per] ? (recv[r * per] + 1) * 8 : 0;
            }
            items.push_back(std::move(it));
        }
        T.exchange_items(items, c->stream);
        CBLX_HIP(hipEventCreateWithFlags(&ev[t], hipEventDisableTiming));
        T.record(ev[t], c->stream);
    }
    Replica rep;
    rep.src.resize(W);
    rep.parts = Q;
    rep.part_ready = ev;
    for (u32 r = 0; r < W; ++r) {
        ReplicaSource& R = rep.src[r];
        R.nseq = recv[r * per];
        for (u32 t = 0; t <= NT; ++t) R.cuts.push_back(hdr_cut(r, t));
        if (r == me) { R.view = ascii_view(d_bases); R.d_off = d_offsets + n0; }
        else {  // planes indexed by the SENDER's base positions: group g of the sender is word g - (its first group) of its share
            R.view = BaseView{nullptr, all_codes.get() + gro[r] - ga_of(r, 0), all_valid.get() + gro[r] - ga_of(r, 0)};
            R.d_off = all_off.get() + sqo[r];
        }
    }
    const bool done = sharded_insert_grouped<C>(c, cm, d_bases, d_offsets, n, cuts, nslices, bounds, false, &rep);
    T.wait();
    CBLX_HIP(hipStreamSynchronize(c->stream));  // the gather buffers die here
    return done;
}

// one rank's FINE plan: the one "group cut" below which the bins are blocks of 2^16 prefixes (insert_device_fine; a one-rank communicator)
inline u32 fine_single_cut(u32 PB) { return std::min<u32>(245u << FINE_LEVEL, (1u << (PB - 1)) - (1u << FINE_LEVEL)); }
template <typename C>
void sharded_insert(cblx_ctx* c, cblx_comm* cm, const u8* d_bases, const u64* d_offsets, u64 n, const u64* cuts, u32 nslices, u32* bounds, int* bounds_valid) {
    Transport& T = *cm->t;
    T.begin_job();
    if (nslices && !*bounds_valid) {
        if (cuts[1] < cuts[0] || cuts[1] > n) throw Error(CBLX_EINVAL, "slice cuts must be ascending and at most n");
        std::vector<u64> hist;
        const u32 asked = cm->protocol == CBLX_PROTO_AUTO ? auto_protocol(T.world) : cm->protocol;
        const bool want_bins = asked == CBLX_PROTO_BINS || asked == CBLX_PROTO_REPLICATE;
        const u32 G = recv_groups_wanted(cm);
        choose_bounds_from_slice<C>(c, T, d_bases, d_offsets + cuts[0], cuts[1] - cuts[0], bounds, &hist, (want_bins && G >= 2) ? G : 0u);
        *bounds_valid = 1;
        if (T.world > 1) {  // the group cuts of the grouped receiver come from the same histogram
            cm->g_cuts = choose_group_cuts(hist, bounds, T.world, recv_groups_wanted(cm), c->P.PB);
            cm->g_bounds.assign(bounds, bounds + (T.world - 1));
        }
    }
    cm->groups_used = 0;
    cm->groups_fine = 0;
    u32 proto = cm->protocol == CBLX_PROTO_AUTO ? auto_protocol(T.world) : cm->protocol;
    if ((proto == CBLX_PROTO_BINS || proto == CBLX_PROTO_REPLICATE) && !bins_protocol_fits(c->P, bounds, T.world)) proto = CBLX_PROTO_SORTED;
    if (proto == CBLX_PROTO_REPLICATE) {
        // (every term of the decision inside is replicated: the ranks take the fallback together)
        cm->protocol_used = CBLX_PROTO_REPLICATE;
        if (T.world >= 2 && sharded_insert_replicate<C>(c, cm, d_bases, d_offsets, n, cuts, nslices, bounds)) return;
        proto = CBLX_PROTO_BINS;  // a non-empty index, a plan the cuts refuse, one rank: the records cross the wire as before
    }
    cm->protocol_used = proto;
    if (proto == CBLX_PROTO_BINS) {
        if (T.world == 1 && c->P.PB > 24 && c->res.count == 0) {
            // a one-rank group at PREFIX_BITS > 24: the FINE-bins build of insert_device_fine on this communicator (its slices as given)
            std::vector<u32> keep = std::move(cm->g_cuts);
            cm->g_cuts.assign(1, fine_single_cut(c->P.PB));
            bool done = false;
            try { done = sharded_insert_grouped<C>(c, cm, d_bases, d_offsets, n, cuts, nslices, bounds, true); } catch (...) { cm->g_cuts = std::move(keep); throw; }
            cm->g_cuts = std::move(keep);
            if (done) { ++c->fine_builds; return; }
        }
        if (sharded_insert_grouped<C>(c, cm, d_bases, d_offsets, n, cuts, nslices, bounds)) return;
        sharded_insert_bins<C>(c, T, ascii_view(d_bases), d_offsets, n, cuts, nslices, bounds, [](u32) {});
    }
    else
        sharded_insert_sorted<C>(c, T, d_bases, d_offsets, n, cuts, nslices, bounds, bounds_valid);
}

// ---- one GPU, a batch that is still arriving: the same slice-by-slice path on a one-rank "group" ---------------------------------
struct LocalTransport : Transport {
    LocalTransport() { rank = 0; world = 1; }
    void all_reduce_sum_u64(u64*, size_t) override {}
    void all_to_all_u64(const u64* send, u64* recv, size_t per) override { std::memcpy(recv, send, per * 8); }
    void exchange(const u8*, const u64*, u8*, const u64*, hipStream_t) override {}
    void wait() override {}
    void exchange_items(const std::vector<Item>&, hipStream_t) override {}
};
// ---- one GPU, PREFIX_BITS > 24, an empty index: the build on FINE bins --------------------------------------------------------------
// The plain build sorts 24 prefix bits in three passes and finishes the last PREFIX_BITS - 24 run by run (k_prefix_split): four trips
// of every record through HBM. Here the first pass runs on the FINE bins of the grouped receiver (cuts.hpp: make_fine_plan) with ONE
// rank and TWO "groups": the prefixes below 245 * 2^16 — where nearly nine tenths of the necklace prefixes of random reads lie — in
// aligned blocks of 2^16 prefixes, whose records need 16 more bits sorted (two passes), and the rest in blocks of 2^24 (three passes of
// 8 bits). No sampling: the cut is fixed, and a batch whose words all lie above it simply takes three passes behind the first, as the
// plain build does. Same index (tests run every PREFIX_BITS > 24 shape through both routes).
bool insert_device_fine(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq) {
    const char* fe = std::getenv("CBLX_FINE_BINS");
    if ((fe && fe[0] == '0') || c->P.PB <= 24 || nseq == 0 || c->res.count != 0) return false;
    check_aligned16(d_bases, "d_bases");
    cblx_comm cm;
    cm.t.reset(new LocalTransport());
    cm.recv_groups = 2;
    cm.g_cuts.assign(1, fine_single_cut(c->P.PB));
    const u64 cuts[2] = {0, nseq};
    bool done = false;
    dispatch(c->P, [&](auto cfg) { done = sharded_insert_grouped<decltype(cfg)>(c, &cm, d_bases, d_offsets, nseq, cuts, 1, nullptr, true); });
    if (done) ++c->fine_builds;
    return done;
}
}  // namespace (reopened below: insert_device_streamed is declared in ingest.hpp)

namespace {
// flush() of a batch sent from pinned host memory in slices (ingest_seqs): KRN-1 and the first partition pass of slice k run
// as soon as the slice has landed, while the later slices are still crossing PCIe; when the last one is in, what is left is
// the remaining passes and the bucket kernels. Same result as insert_device (the slices are pieces of the pass-A segments in
// stream order, exactly what the receiver of the multi-GPU build gets from its senders).
// `ready(s)`: returns once the ctx's stream may read slice s (s = ~0u: the offsets) — it makes the stream wait on the transfer's
// events, blocking the host first if they are not recorded yet
template <typename Ready>
void insert_device_sliced(cblx_ctx* c, const BaseView& bases, const u64* d_offsets, u64 nseq, const std::vector<u64>& seq_cuts, Ready&& ready) {
    const u32 ns = (u32)seq_cuts.size() - 1;
    {   // PREFIX_BITS > 24 on an empty index: the FINE-bins build (insert_device_fine's plan: one rank, two groups, a fixed cut) slice by slice as
        // the batch lands — two LSD passes behind the first one for the dense part of the prefix space instead of three for everything (round 6)
        const char* fe = std::getenv("CBLX_FINE_BINS");
        if (c->P.PB > 24 && c->res.count == 0 && nseq && ns && !(fe && fe[0] == '0')) {
            cblx_comm cm;
            cm.t.reset(new LocalTransport());
            cm.recv_groups = 2;
            cm.g_cuts.assign(1, fine_single_cut(c->P.PB));
            Replica rep;
            rep.src.resize(1);
            rep.src[0].view = bases;
            rep.src[0].d_off = d_offsets + seq_cuts[0];
            rep.src[0].nseq = seq_cuts[ns] - seq_cuts[0];
            for (u64 x : seq_cuts) rep.src[0].cuts.push_back(x - seq_cuts[0]);
            rep.per_slice = true;
            rep.ready = [&](u32 s) { ready(s); };
            bool done = false;
            dispatch(c->P, [&](auto cfg) { done = sharded_insert_grouped<decltype(cfg)>(c, &cm, nullptr, d_offsets, nseq, seq_cuts.data(), ns, nullptr, true, &rep); });
            if (done) { ++c->fine_builds; collect_events(c); return; }
        }
    }
    LocalTransport T;
    u32 none = 0;
    dispatch(c->P, [&](auto cfg) { sharded_insert_bins<decltype(cfg)>(c, T, bases, d_offsets, nseq, seq_cuts.data(), ns, &none, ready); });
    collect_events(c);
}
void insert_device_streamed(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, const Ingest::Streamed& plan) {
    auto wait_for = [&](const std::vector<hipEvent_t>& evs) { for (hipEvent_t e : evs) CBLX_HIP(hipStreamWaitEvent(c->stream, e, 0)); };
    if (c->P.PB < 9 || nseq == 0) {  // no LSD pass behind pass A: the plain path, once everything is there
        wait_for(plan.offsets_ready);
        for (auto& v : plan.ready) wait_for(v);
        insert_device(c, d_bases, d_offsets, nseq);
        return;
    }
    check_aligned16(d_bases, "d_bases");
    const bool trace = std::getenv("CBLX_TRACE_H2D") != nullptr;  // dev: when every slice had landed / was handed to the kernels
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    insert_device_sliced(c, ascii_view(d_bases), d_offsets, nseq, plan.seq_cuts, [&](u32 s) {
        if (s == ~0u) { wait_for(plan.offsets_ready); return; }
        if (trace) {
            const double a = ms();
            for (hipEvent_t e : plan.ready[s]) CBLX_HIP(hipEventSynchronize(e));
            fprintf(stderr, "[cblx h2d] slice %u: host arrives %.2f ms, landed %.2f ms\n", s, a, ms());
        }
        wait_for(plan.ready[s]);
    });
    if (trace) { CBLX_HIP(hipStreamSynchronize(c->stream)); fprintf(stderr, "[cblx h2d] done %.2f ms\n", ms()); }
}

// A big host batch as BIT PLANES (3 bits per base over PCIe instead of 8: the link is the bound of the host-input path). Host
// threads pack the caller's ASCII bases unit by unit (xfer.hpp: pack_planes) into pinned staging, an issuer thread copies every
// finished unit to the device on two streams, and THIS thread runs the sliced insert right behind them: slice s = the sequences
// that end inside the units landed so far. The index is built when the call returns (the caller's buffers are borrowed for the call
// only). Pageable or pinned source alike. Returns false when the batch does not qualify (the caller takes the other paths).
// check_slice(i0, i1): throws unless sequences [i0, i1) of the batch are well-formed (offsets non-decreasing, every length >= K)
template <typename V> bool ingest_seqs_planes(cblx_ctx* c, const u8* bases, const u64* offsets, u64 n, V&& check_slice) {
    Ingest& g = c->ing;
    const bool trace = std::getenv("CBLX_TRACE_H2D") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const u64 o0 = offsets[0], len = offsets[n] >= o0 ? offsets[n] - o0 : 0;
    const char* pk = std::getenv("CBLX_H2D_PACK");  // 0 off, 1 on, default: by core count (read per call: tests switch it)
    const int mode = pk ? std::atoi(pk) : -1;
    const unsigned hc = std::thread::hardware_concurrency();
    if (mode == 0 || (mode < 0 && hc < 16)) return false;
    if (g.nseq || g.nbytes || g.query || g.staged || c->P.PB < 9 || len < (64u << 20) || n < 1024 || len >= ingest_flush_bytes()) return false;
    // transfer units = slices of the insert. Few: every slice costs fixed work (chunk plan, column scans, piece: ~0.5 ms). Measured at
    // cfg 2 with equal units: 4 / 6 / 8 / 12 / 16 / 32 units 38.8 / 37.6 / 37.8 / 39.2 / 40.5 / 46.5 ms. The kernel time line of that
    // (tools/dev_h2d_timeline.py): the wire is the bound while the units arrive (0.64 GB = 11.7 ms; KRN-1 + pass A of all of them
    // take 10.2), in front of it the packing of the first unit, behind it the kernels of the last — so the six default units
    // are 1, 3, 4, 4, 3, 1 sixteenths of the batch: a short first one (the wire starts early) and a short last one.
    const char* eu = std::getenv("CBLX_H2D_UNITS");
    const u32 NU = eu && std::atoi(eu) > 0 ? (u32)std::min(std::atoi(eu), 64) : 6u;
    std::vector<u64> ub(1, 0);  // unit k = bases [ub[k], ub[k + 1]), multiples of 1024
    {
        std::vector<u32> w;  // relative sizes (CBLX_H2D_TAPER="1,3,4,4,3,1" overrides; CBLX_H2D_UNITS = that many equal units)
        if (const char* et = std::getenv("CBLX_H2D_TAPER")) {
            for (const char* p = et; *p;) { char* q; const unsigned long v = std::strtoul(p, &q, 10); if (q == p) break; if (v) w.push_back((u32)v); p = *q ? q + 1 : q; }
        }
        if (w.empty()) { if (eu) w.assign(NU, 1u); else w = {1, 3, 4, 4, 3, 1}; }
        u64 tot = 0, acc = 0;
        for (u32 v : w) tot += v;
        for (size_t k = 0; k < w.size(); ++k) {
            acc += w[k];
            const u64 b = k + 1 == w.size() ? len : std::min<u64>(len, ((u64)((double)len * (double)acc / (double)tot) + 1023) & ~(u64)1023);
            if (b > ub.back()) ub.push_back(b);
        }
        if (ub.back() < len) ub.push_back(len);
    }
    const u32 nu = (u32)ub.size() - 1;
    auto unit_at = [&](u64 base) -> u32 { return (u32)(std::upper_bound(ub.begin(), ub.end(), base) - ub.begin()) - 1u; };
    const u64 ng = (len + 15) / 16;                                      // groups of 16 bases
    // device planes, offsets, pinned staging (kept between calls)
    if (g.d_codes.n < ng + 8) g.d_codes = Buf<u32>(c->pool, ng + 8);
    if (g.d_valid.n < ng + 8) g.d_valid = Buf<u16>(c->pool, ng + 8);
    ingest_reserve(c, 0, n);
    const size_t need = (size_t)(ng + 8) * 6;
    if (g.pin_cap < need) {
        if (g.pin) { u8* old = g.pin; g.pin = nullptr; g.pin_cap = 0; CBLX_HIP(hipHostFree(old)); }
        CBLX_HIP(hipHostMalloc((void**)&g.pin, need, hipHostMallocDefault));
        g.pin_cap = need;
    }
    u32* h_codes = (u32*)g.pin;
    u16* h_valid = (u16*)(g.pin + (size_t)(ng + 8) * 4);
    Xfer& x = xfer(c);
    hipStream_t cs[2] = {x.lane_stream(0), x.lane_stream(1)};
    // offsets (relative to the batch's first base): straight from the caller's array when pinned and zero-based, else transformed
    hipEvent_t off_ev = nullptr;
    // pinned and zero-based: the offsets of a slice travel in front of its unit (all of them up front are 80 MB at cfg 2 — 1.4 ms of
    // wire before the first base); only the last one, which sizes the arena, goes ahead
    const bool off_by_slice = o0 == 0 && Xfer::is_pinned(offsets) && Xfer::is_pinned(offsets + n);
    if (off_by_slice) x.h2d_pinned_lane0(g.d_off.get() + n, offsets + n, 8, off_ev);
    else {
        x.h2d(g.d_off.get() + 1, n * 8, [&](u8* dst, size_t off, size_t nb) {
            u64* d = (u64*)dst;
            const u64* src = offsets + off / 8 + 1;
            for (size_t j = 0; j < nb / 8; ++j) d[j] = src[j] - o0;
        });
        x.sync();
    }
    // slices: slice k = the sequences that end inside units 0 .. k
    std::vector<u64> cuts(1, 0);
    std::vector<u32> unit_of;  // the last unit a slice needs
    for (u32 k = 0; k < nu; ++k) {
        const u64 lim = ub[k + 1];
        const u64 i = k + 1 == nu ? n : (u64)(std::upper_bound(offsets + 1, offsets + n + 1, o0 + lim) - (offsets + 1));
        if (i > cuts.back()) { cuts.push_back(i); unit_of.push_back(k); }
    }
    // packers: sub-blocks of a unit in order, so that the units complete one after the other
    const u64 SB = 2u << 20;  // bases per sub-block (a multiple of 16)
    const u64 nsb = (len + SB - 1) / SB;
    std::vector<std::atomic<u32>> left(nu);
    for (u32 k = 0; k < nu; ++k) {
        const u64 a = ub[k], b = ub[k + 1];
        left[k].store((u32)((b + SB - 1) / SB - a / SB) , std::memory_order_relaxed);
    }
    // (a unit boundary is a multiple of SB only by accident: a sub-block may straddle units — it then counts for each of them)
    std::vector<std::atomic<u8>> dirty(nu);  // a base outside ACGTacgt in the unit: only then its validity plane is sent
    for (u32 k = 0; k < nu; ++k) dirty[k].store(0, std::memory_order_relaxed);
    std::atomic<u64> next_sb{0};
    std::atomic<u32> issued{0};
    std::atomic<bool> failed{false};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<hipEvent_t> ev(nu, nullptr);
    const int T = (int)std::max(2u, std::min(32u, hc / 2));
    auto pack_worker = [&] {
        for (u64 sb; (sb = next_sb.fetch_add(1)) < nsb;) {
            const u64 a = sb * SB, b = std::min<u64>(len, a + SB);
            const bool clean = pack_planes(bases + o0 + a, b - a, h_codes + a / 16, h_valid + a / 16);
            for (u32 k = unit_at(a), k1 = unit_at(b - 1); k <= k1; ++k) {
                if (!clean) dirty[k].store(1, std::memory_order_relaxed);
                if (left[k].fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> l(mu); cv.notify_all(); }
            }
        }
    };
    auto issuer = [&] {
        try {
            CBLX_HIP(hipSetDevice(c->device));
            for (u32 k = 0; k < nu; ++k) {
                { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return left[k].load(std::memory_order_acquire) == 0; }); }
                const u64 g0 = ub[k] / 16, g1 = k + 1 == nu ? ng : ub[k + 1] / 16;
                if (off_by_slice)
                    for (size_t sl = 0; sl < unit_of.size(); ++sl)
                        if (unit_of[sl] == k && cuts[sl + 1] > cuts[sl])
                            CBLX_HIP(hipMemcpyAsync(g.d_off.get() + 1 + cuts[sl], offsets + 1 + cuts[sl], (cuts[sl + 1] - cuts[sl]) * 8, hipMemcpyHostToDevice, cs[k & 1]));
                CBLX_HIP(hipMemcpyAsync(g.d_codes.get() + g0, h_codes + g0, (g1 - g0) * 4, hipMemcpyHostToDevice, cs[k & 1]));
                if (dirty[k].load(std::memory_order_relaxed)) {
                    CBLX_HIP(hipMemcpyAsync(g.d_valid.get() + g0, h_valid + g0, (g1 - g0) * 2, hipMemcpyHostToDevice, cs[k & 1]));
                } else {
                    // every base of the unit is valid: its validity plane is all ones and is filled in on the device (a third of
                    // the unit's bytes stays off the link); the batch's last, partly filled word comes from the host
                    CBLX_HIP(hipMemsetAsync(g.d_valid.get() + g0, 0xFF, (g1 - g0) * 2, cs[k & 1]));
                    if (k + 1 == nu && (len & 15)) CBLX_HIP(hipMemcpyAsync(g.d_valid.get() + ng - 1, h_valid + ng - 1, 2, hipMemcpyHostToDevice, cs[k & 1]));
                }
                hipEvent_t e;
                CBLX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                CBLX_HIP(hipEventRecord(e, cs[k & 1]));
                ev[k] = e;
                { std::lock_guard<std::mutex> l(mu); issued.store(k + 1, std::memory_order_release); }
                cv.notify_all();
            }
        } catch (...) {
            { std::lock_guard<std::mutex> l(mu); failed = true; issued.store(nu, std::memory_order_release); }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(pack_worker);
    th.emplace_back(issuer);
    struct Join {
        std::vector<std::thread>& th; std::vector<hipEvent_t>& ev; hipEvent_t& off_ev; Xfer& x;
        ~Join() {
            for (auto& t : th) if (t.joinable()) t.join();
            try { x.sync(); } catch (...) {}
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            if (off_ev) (void)hipEventDestroy(off_ev);
        }
    } join{th, ev, off_ev, x};
    if (trace) fprintf(stderr, "[cblx h2d planes] threads started %.2f ms\n", ms());
    const BaseView view{nullptr, g.d_codes.get(), g.d_valid.get()};
    insert_device_sliced(c, view, g.d_off.get(), n, cuts, [&](u32 s) {
        if (s == ~0u) { if (off_ev) CBLX_HIP(hipStreamWaitEvent(c->stream, off_ev, 0)); return; }
        // The slice's offsets are checked before a kernel reads them (the resident index is not touched before the last slice is
        // in: a failure here leaves it as it was), while its unit is on its way and the slices in front of it run.
        check_slice(cuts[s], cuts[s + 1]);
        const u32 k = unit_of[s];
        const double a = trace ? ms() : 0;
        { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return issued.load(std::memory_order_acquire) > k; }); }
        if (failed) throw Error(CBLX_EDEVICE, "host-to-device transfer of the batch failed");
        CBLX_HIP(hipStreamWaitEvent(c->stream, ev[k], 0));               // units are copied in order on two streams:
        if (k) CBLX_HIP(hipStreamWaitEvent(c->stream, ev[k - 1], 0));    // the last one on each of them
        if (trace) fprintf(stderr, "[cblx h2d planes] slice %u (unit %u): host arrives %.2f ms, issued %.2f ms\n", s, k, a, ms());
    });
    if (trace) { CBLX_HIP(hipStreamSynchronize(c->stream)); fprintf(stderr, "[cblx h2d planes] done %.2f ms\n", ms()); }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    return true;
}

// ---- a plain FASTA / FASTQ file as bit planes: the parser threads pack the sequence lines themselves -------------------------------
// The file path was bound by its host side: every region's sequence lines were copied into pinned slots and sent as ASCII (1.5 GB
// for cfg 2, 65 - 97 ms), then inserted window by window. Here a region's thread appends its lines to the batch's bit planes IN
// PLACE (PlaneSink: 16 bytes -> three 16-bit words by SSE2 movemask, shifted to the region's bit offset; only the first and the last
// word of a region are shared with its neighbours and take an atomic OR), the slices (groups of consecutive regions) are copied as
// they complete, and the calling thread runs the sliced insert right behind them, as ingest_seqs_planes does for a host batch.
bool fastx_parallel_planes(cblx_ctx* c, const char* path, u64* nrec_out) {
    Ingest& g = c->ing;
    const char* pk = std::getenv("CBLX_H2D_PACK");
    const int mode = pk ? std::atoi(pk) : -1;
    const unsigned hc = std::thread::hardware_concurrency();
    if (mode == 0 || (mode < 0 && hc < 16)) return false;
    if (g.nseq || g.nbytes || g.query || g.staged || c->P.PB < 9) return false;
    const size_t MIN_BYTES = fastx_env_bytes("CBLX_FASTX_PARALLEL_MIN", 32u << 20), REGION = fastx_env_bytes("CBLX_FASTX_REGION_BYTES", 16u << 20);
    const bool trace = std::getenv("CBLX_INGEST_TRACE") != nullptr;
    const auto t00 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t00).count(); };
    FastxMap m;
    if (!m.open(path, MIN_BYTES)) return false;
    std::vector<FastxRegion> regs;
    fx_make_regions(m, m.first, m.size, REGION, regs);
    if (trace) fprintf(stderr, "[fastx planes] mapped, %zu regions at %.2f ms\n", regs.size(), ms());
    const unsigned TP = (unsigned)fastx_env_bytes("CBLX_FASTX_THREADS", std::min(32u, std::max(2u, hc / 2)));  // parser threads (both passes)
    if (!fx_count_regions(m, regs, c->P.K, TP)) return false;
    if (trace) fprintf(stderr, "[fastx planes] counted at %.2f ms\n", ms());
    const u32 K = c->P.K;
    const u64 window = std::min<u64>(ingest_flush_bytes(), 0x78000000ull);  // bases per batch (one batch = fewer than 2^32 words)
    u64 total_rec = 0;
    Xfer& x = xfer(c);
    hipStream_t cs[2] = {x.lane_stream(0), x.lane_stream(1)};
    for (size_t w0 = 0; w0 < regs.size();) {
        size_t w1 = w0;
        u64 nbases = 0, nrec = 0;
        while (w1 < regs.size() && (w1 == w0 || nbases + regs[w1].nbases <= window)) { nbases += regs[w1].nbases; nrec += regs[w1].nrec; ++w1; }
        const size_t nr = w1 - w0;
        if (nrec == 0) { w0 = w1; continue; }
        const u64 ng = (nbases + 15) / 16;
        if (g.d_codes.n < ng + 8) g.d_codes = Buf<u32>(c->pool, ng + 8);
        if (g.d_valid.n < ng + 8) g.d_valid = Buf<u16>(c->pool, ng + 8);
        ingest_reserve(c, 0, nrec);
        const size_t need = (size_t)(ng + 8) * 6 + (size_t)nrec * 8 + 64;
        if (g.pin_cap < need) {
            if (g.pin) { u8* old = g.pin; g.pin = nullptr; g.pin_cap = 0; CBLX_HIP(hipHostFree(old)); }
            CBLX_HIP(hipHostMalloc((void**)&g.pin, need, hipHostMallocDefault));
            g.pin_cap = need;
        }
        u32* h_codes = (u32*)g.pin;
        u16* h_valid = (u16*)(g.pin + (size_t)(ng + 8) * 4);
        u64* h_ends = (u64*)(g.pin + (((size_t)(ng + 8) * 6 + 63) & ~(size_t)63));
        // the last offset (= the window's bases, known from the count) goes ahead of everything: the sliced insert sizes its arena
        // from it before the first slice (the offsets themselves travel with their slices)
        CBLX_HIP(hipMemcpy(g.d_off.get() + nrec, &nbases, 8, hipMemcpyHostToDevice));
        std::vector<u64> base(nr + 1, 0), rec0(nr + 1, 0);
        for (size_t i = 0; i < nr; ++i) { base[i + 1] = base[i] + regs[w0 + i].nbases; rec0[i + 1] = rec0[i] + regs[w0 + i].nrec; }
        fx_planes_prezero(base, ng + 8, h_codes, h_valid);
        // slices = groups of consecutive regions of about 1 / NS of the bases
        const u32 NS = 6;
        std::vector<size_t> sl(1, 0);
        for (u32 k = 1; k <= NS; ++k) {
            size_t i = sl.back();
            while (i < nr && base[i] < nbases * k / NS) ++i;
            if (k == NS) i = nr;
            if (i > sl.back()) sl.push_back(i);
        }
        const u32 ns = (u32)sl.size() - 1;
        std::vector<u64> cuts(ns + 1);
        for (u32 k = 0; k <= ns; ++k) cuts[k] = rec0[sl[k]];
        std::vector<u32> slice_of(nr);
        std::vector<std::atomic<u32>> left(ns);
        for (u32 k = 0; k < ns; ++k) { left[k].store((u32)(sl[k + 1] - sl[k])); for (size_t i = sl[k]; i < sl[k + 1]; ++i) slice_of[i] = k; }
        std::atomic<size_t> next{0};
        std::atomic<u32> issued{0};
        std::atomic<bool> failed{false};
        std::mutex mu;
        std::condition_variable cv;
        std::vector<hipEvent_t> ev(ns, nullptr);
        const u8* d = m.d;
        const char fmt = m.fmt;
        auto worker = [&] {
            for (size_t i; (i = next.fetch_add(1)) < nr;) {
                const FastxRegion& r = regs[w0 + i];
                PlaneSink sink(h_codes, h_valid, base[i], h_ends + rec0[i], r.nrec, base[i + 1]);
                if (!fx_walk(d, r, fmt, K, sink) || sink.overflow || sink.pos != base[i + 1] || sink.nrec != r.nrec) failed = true;
                sink.finish();
                if (left[slice_of[i]].fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> l(mu); cv.notify_all(); }
            }
        };
        auto issuer = [&] {
            try {
                CBLX_HIP(hipSetDevice(c->device));
                for (u32 k = 0; k < ns; ++k) {
                    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return left[k].load(std::memory_order_acquire) == 0; }); }
                    // the slice's groups, its last, possibly shared word included. A slice that STARTS inside a group shares that
                    // word with its predecessor, whose copy of it (taken before this slice's bits were all there) runs on the other
                    // stream: two DMA writes of one word with nothing between them — whichever lands last stays. So the word this
                    // slice shares with its predecessor is copied separately, BEHIND the predecessor's event; the bulk does not wait.
                    const bool shared_front = k > 0 && (base[sl[k]] & 15) != 0;
                    const u64 g0 = base[sl[k]] >> 4, g1 = std::min<u64>(ng, (base[sl[k + 1]] + 15) >> 4), gb = std::min(g1, g0 + (shared_front ? 1 : 0));
                    hipStream_t st = cs[k & 1];
                    if (g1 > gb) {
                        CBLX_HIP(hipMemcpyAsync(g.d_codes.get() + gb, h_codes + gb, (g1 - gb) * 4, hipMemcpyHostToDevice, st));
                        CBLX_HIP(hipMemcpyAsync(g.d_valid.get() + gb, h_valid + gb, (g1 - gb) * 2, hipMemcpyHostToDevice, st));
                    }
                    if (gb > g0) {
                        CBLX_HIP(hipStreamWaitEvent(st, ev[k - 1], 0));
                        CBLX_HIP(hipMemcpyAsync(g.d_codes.get() + g0, h_codes + g0, 4, hipMemcpyHostToDevice, st));
                        CBLX_HIP(hipMemcpyAsync(g.d_valid.get() + g0, h_valid + g0, 2, hipMemcpyHostToDevice, st));
                    }
                    if (cuts[k + 1] > cuts[k]) CBLX_HIP(hipMemcpyAsync(g.d_off.get() + 1 + cuts[k], h_ends + cuts[k], (cuts[k + 1] - cuts[k]) * 8, hipMemcpyHostToDevice, st));
                    hipEvent_t e;
                    CBLX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    CBLX_HIP(hipEventRecord(e, st));
                    ev[k] = e;
                    { std::lock_guard<std::mutex> l(mu); issued.store(k + 1, std::memory_order_release); }
                    cv.notify_all();
                }
            } catch (...) {
                { std::lock_guard<std::mutex> l(mu); failed = true; issued.store(ns, std::memory_order_release); }
                cv.notify_all();
            }
        };
        const int T = (int)std::max<size_t>(1, std::min<size_t>(TP, nr));
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(worker);
        th.emplace_back(issuer);
        struct Join {
            std::vector<std::thread>& th; std::vector<hipEvent_t>& ev; Xfer& x;
            ~Join() {
                for (auto& t : th) if (t.joinable()) t.join();
                try { x.sync(); } catch (...) {}
                for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            }
        } join{th, ev, x};
        const BaseView view{nullptr, g.d_codes.get(), g.d_valid.get()};
        insert_device_sliced(c, view, g.d_off.get(), nrec, cuts, [&](u32 s) {
            if (s == ~0u) s = 0;  // the offsets of a slice travel with it: the first read needs slice 0
            { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return issued.load(std::memory_order_acquire) > s; }); }
            if (failed) throw Error(CBLX_EDEVICE, "fastx: the file changed while it was being read");
            CBLX_HIP(hipStreamWaitEvent(c->stream, ev[s], 0));
            if (s) CBLX_HIP(hipStreamWaitEvent(c->stream, ev[s - 1], 0));
            if (trace) fprintf(stderr, "[fastx planes] slice %u handed over at %.2f ms\n", s, ms());
        });
        CBLX_HIP(hipStreamSynchronize(c->stream));
        if (failed) throw Error(CBLX_EDEVICE, "fastx: the file changed while it was being read");
        total_rec += nrec;
        if (trace) fprintf(stderr, "[fastx planes] window of %llu records done at %.2f ms\n", (unsigned long long)nrec, ms());
        w0 = w1;
    }
    if (nrec_out) *nrec_out = total_rec;
    return true;
}

}  // namespace
