/* cblx.h — C ABI of the MI355X-native bulk k-mer insertion path for CBL indexes.
 *
 * One `cblx_ctx` stands for one `CBL<K, T, PREFIX_BITS>` value of the reference
 * (/root/reference/src/cbl.rs:40-54). The reference has no plugin registry; its only FFI is
 * autocxx -> C++ for RankBV / TieredVec32 (/root/reference/src/ffi.rs:7-20). This header is the
 * `extern "C"` layer a Rust `CBL<K,T,PREFIX_BITS>` facade binds instead:
 * const generics become the runtime fields of `cblx_params`; `panic!` becomes a non-zero status +
 * `cblx_last_error` (the shim re-panics with the same message).
 *
 * Conventions
 *   - every function returns 0 on success, a CBLX_E* code otherwise; nothing throws across the ABI;
 *   - plain pointers + sizes only; input pointers are borrowed for the duration of the call;
 *   - `*_device` entry points take DEVICE pointers valid on the ctx's GPU, all others HOST pointers;
 *   - a ctx is single-owner / not re-entrant (the reference API is `&mut self` throughout);
 *   - `cblx_insert_seq*` may only enqueue; every observer flushes first.
 *   - the HIP library is the only implementation: there is no CPU fallback behind this ABI.
 */
#ifndef CBLX_H
#define CBLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: cblx_merge_from, cblx_stage_units, cblx_fine_builds, cblx_comm_groups_fine, cblx_comm_protocol_used, CBLX_PROTO_AUTO (the default of a new
 * communicator: an unchanged 2 - 4 rank caller no longer runs BINS), CBLX_PROTO_REPLICATE; empty PREFIX_BITS > 24 builds take the FINE route. A binding
 * built against this header must refuse a library that reports less. */
#define CBLX_ABI_VERSION 3

enum {
    CBLX_OK = 0,
    CBLX_EINVAL = 1,   /* bad parameter (K, PREFIX_BITS, null pointer, params mismatch) */
    CBLX_ESHORT = 2,   /* "Sequence size (n) is smaller than K (k)"   src/cbl.rs:329-334 */
    CBLX_EFORMAT = 3,  /* index bytes do not parse / trailing bytes    src/cbl.rs:146-158 */
    CBLX_EDEVICE = 4,  /* HIP runtime error or no usable GPU */
    CBLX_ENOMEM = 5,
    CBLX_ERANGE = 6    /* output buffer too small */
};

typedef struct cblx_ctx cblx_ctx;

typedef struct cblx_params {
    uint32_t k;           /* const K: odd, 5..59 (build.rs:18-24; 2K + POS_BITS <= 128, src/cbl.rs:87-91) */
    uint32_t prefix_bits; /* const PREFIX_BITS: 1..32 and < 2K + POS_BITS (src/wordset/mod.rs:37-41); default 24 */
    uint32_t canonical;   /* CBL::new() = 0, CBL::new_canonical() = 1 (src/cbl.rs:71-79) */
    int32_t device;       /* HIP device ordinal; -1 = current device */
    uint32_t flags;       /* CBLX_FLAG_* */
    uint32_t reserved;
} cblx_params;

#define CBLX_FLAG_PROFILE 1u /* record per-stage HIP-event timings of each flush (cblx_stage_times) */

uint32_t cblx_abi_version(void);
/* Error text of the last failed call on this thread that had no ctx (cblx_create). */
const char* cblx_last_global_error(void);

/* CBL::new / new_canonical (src/cbl.rs:71-79, asserts :87-91 and src/wordset/mod.rs:37-41). */
int cblx_create(const cblx_params* params, cblx_ctx** out);
/* Drop (frees device memory owned by the ctx). */
void cblx_destroy(cblx_ctx* ctx);
const char* cblx_last_error(const cblx_ctx* ctx);

/* CBL::insert_seq (src/cbl.rs:328-339): enqueue ONE sequence of ASCII nucleotides (copied). */
int cblx_insert_seq(cblx_ctx* ctx, const uint8_t* seq, uint64_t len);
/* The caller loop `for record in reader { cbl.insert_seq(record.seq()) }` (examples/cbl.rs:160-163,242-245)
 * in one call: sequence i is bases[offsets[i] .. offsets[i+1]), i < n. Stream order = i ascending. */
int cblx_insert_seqs(cblx_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n);
/* Same, inputs already resident in HBM (device pointers; d_bases 16-byte aligned). offsets[0] need not be 0: a slice
 * offsets[a..b] of a larger batch addresses the same d_bases. Runs the insert before returning. */
int cblx_insert_seqs_device(cblx_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n);
/* The reader loop of examples/cbl.rs:154-163 (needletail stand-in): every record of a plain-text FASTA (multi-line ok)
 * or 4-line FASTQ file, plain or gzip (zlib is looked up at run time), goes through insert_seq, in file order. */
int cblx_insert_fastx_file(cblx_ctx* ctx, const char* path, uint64_t* n_records);
/* Materialise everything enqueued so far into the resident index (idempotent). */
int cblx_flush(cblx_ctx* ctx);

/* The reader loop of examples/cbl.rs:154-163 for a multi-GPU build from ONE file with file-order parity: parses the file like
 * cblx_insert_fastx_file but INSERTS NOTHING; the records that block-cyclic dealing gives this rank — record i (file order,
 * 0-based) belongs to rank (i / block) % world — are staged in HBM, in file order, and lent to the caller as device arrays
 * (bases, n_staged + 1 offsets, d_offsets[0] = 0; d_bases 16-byte aligned) until cblx_stage_release. Local block c (records
 * [c * block, (c + 1) * block) of the staged list) is block c * world + rank of the file, so a sharded build that feeds local
 * block c as slice c has the stream order of the file. *n_in_file = records in the whole file. block = 0: count only.
 * While records are staged the ctx takes no cblx_insert_seq* calls; its index can be read and inserted into by the
 * *_device entry points. A record shorter than K is CBLX_ESHORT on the rank that owns it. */
int cblx_stage_fastx_blocks(cblx_ctx* ctx, const char* path, uint64_t block, uint32_t rank, uint32_t world, const uint8_t** d_bases,
                            const uint64_t** d_offsets, uint64_t* n_staged, uint64_t* n_in_file);
int cblx_stage_release(cblx_ctx* ctx);
/* The same staging for the ranks of a communicator (rank / world are the communicator's), with the parse SHARED between them:
 * rank r counts the records of bytes [r, r + 1) * size / world of the file, the counts and the byte offsets of the block starts
 * are summed over the ranks, and every rank then reads only the bytes of ITS blocks — host time per rank ~ 1 / world instead of
 * one whole-file parse per rank. *block: records per block; 0 = chosen here so that every rank gets about `slices` blocks
 * (written back). A file the parallel readers do not take (gzip, a record shorter than K, a malformed FASTQ record) falls back
 * to cblx_stage_fastx_blocks on every rank. Collective: every rank of the communicator calls it with the same path. */
struct cblx_comm;
int cblx_stage_fastx_blocks_comm(cblx_ctx* ctx, struct cblx_comm* comm, const char* path, uint64_t* block, uint32_t slices,
                                 const uint8_t** d_bases, const uint64_t** d_offsets, uint64_t* n_staged, uint64_t* n_in_file);

/* Device word arrays (all *_words_device entry points): word i = (hi[i] << 64) | lo[i]. `lo` is uint64_t[]; the element
 * type of `hi` follows from K like the reference's T (build.rs:34-41) and is reported by cblx_consts.hi_bytes:
 *   0 -> no hi array (2K + POS_BITS <= 64; pass NULL), 1 -> uint8_t[] (K = 31: 68-bit words), 8 -> uint64_t[] (K >= 33). */

/* WordSet::insert_batch (src/wordset/mod.rs:187-216) on already transformed words, stream order = index order.
 * On an empty index the first partition pass reads the caller's arrays in place (no copy). */
int cblx_insert_words_device(cblx_ctx* ctx, const uint64_t* d_lo, const void* d_hi, uint64_t n);
/* CBL::get_seq_words over every chunk of every sequence (src/cbl.rs:239-289), no insertion: writes the
 * words in stream order to d_lo/d_hi (device, capacity `cap` words) and their number to *n_words. */
int cblx_seq_words_device(cblx_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n,
                          uint64_t* d_lo, void* d_hi, uint64_t cap, uint64_t* n_words);

/* Multi-GPU exchange step (no reference counterpart; the reference is single-process): STABLE partition of n words
 * by destination = #{i : bounds[i] <= prefix(word)}, nd destinations (<= 16), bounds[nd-1] ascending prefix values
 * (host). Destination d's words land contiguously, in input order, at d_out[sum(counts[0..d])..]; counts[nd] (host)
 * receives the run lengths. Device word arrays as above. */
int cblx_partition_words_device(cblx_ctx* ctx, const uint64_t* d_lo, const void* d_hi, uint64_t n, const uint32_t* bounds,
                                uint32_t nd, uint64_t* d_out_lo, void* d_out_hi, uint64_t* counts);

/* cblx_seq_words_device + cblx_partition_words_device in one call (the per-slice step of the multi-GPU build): the
 * words of the sequences, grouped by destination, written to d_out_lo/d_out_hi (capacity `cap` words). */
int cblx_seq_words_partitioned_device(cblx_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n,
                                      const uint32_t* bounds, uint32_t nd, uint64_t* d_out_lo, void* d_out_hi, uint64_t cap,
                                      uint64_t* counts, uint64_t* n_words);

/* Multi-GPU build, "sorted batch" protocol (no reference counterpart). A rank partitions ITS words completely before the
 * exchange; per destination it ships a slice of the prefix-sorted batch — the non-empty prefixes, their word counts and
 * the suffixes alone, packed to suffix_bytes (= cblx_consts.bytes) little-endian bytes each: 6 B per word at K=31 /
 * PREFIX_BITS=24 instead of the 9 B of a full word — and the receiver merges the batches of all ranks bucket by bucket
 * without partitioning anything again. */
typedef struct cblx_batch_view {
    uint64_t n_buckets;       /* non-empty prefixes of the batch */
    uint64_t n_words;
    const uint32_t* d_prefix; /* device [n_buckets], strictly ascending */
    const uint32_t* d_count;  /* device [n_buckets], words per prefix */
    const uint8_t* d_suffix;  /* device [n_words * suffix_bytes], bucket-major, stream order inside a bucket */
} cblx_batch_view;
/* Sender: KRN-1 + the full stable partition of the words of n device-resident sequences. The batch stays in the ctx until
 * it is exported (or replaced by the next begin). Destination d = #{i : bounds[i] <= prefix} (nd destinations, nd-1
 * ascending host bounds) owns buckets [bucket_split[d], bucket_split[d+1]) and words [word_split[d], word_split[d+1])
 * of it (host arrays of nd + 1 entries). */
int cblx_sorted_batch_begin(cblx_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, const uint32_t* bounds,
                            uint32_t nd, uint64_t* bucket_split, uint64_t* word_split);
/* Writes the pending batch — all destinations, in order — to caller-owned device arrays of bucket_split[nd],
 * bucket_split[nd] and word_split[nd] * suffix_bytes elements, and drops it. */
int cblx_sorted_batch_export(cblx_ctx* ctx, uint32_t* d_prefix, uint32_t* d_count, uint8_t* d_suffix);
/* Receiver: WordSet::insert_batch (src/wordset/mod.rs:187-216) over n_batches sorted batches; stream order = batch
 * order (then bucket order is irrelevant, and stream order inside a bucket of a batch is kept). */
int cblx_insert_sorted_batches_device(cblx_ctx* ctx, const cblx_batch_view* batches, uint32_t n_batches);

/* ---- prefix-range sharded indexes (BASELINE.json cfg 5 on N GPUs; no reference counterpart: the reference is one process) ----
 * `cbl merge a b -o out` (examples/cbl.rs:270-279) per prefix range: every rank loads ITS range of both files
 * (read_index, examples/cbl.rs:117-130), merges (src/wordset/set_ops.rs:123-157: buckets are independent by prefix) and
 * writes its entries at its own offset of the output (write_index, examples/cbl.rs:132-142). Two operands cut at
 * different bounds are brought to common bounds by moving bucket batches between ranks (export -> exchange -> install). */
typedef struct cblx_shard_info {
    uint64_t header_entries;   /* entry count in the file header (the whole index) */
    uint64_t local_entries;    /* entries this rank loaded */
    uint64_t begin_off, end_off; /* byte range of the file this rank parsed */
    uint32_t first_prefix, last_prefix; /* of the loaded entries (when local_entries != 0) */
    uint32_t exact;            /* 1: the walk over [begin_off, end_off) ended exactly at end_off and nothing looked wrong */
    uint32_t canonical;        /* the file's canonical flag */
} cblx_shard_info;
/* Replaces the ctx's contents by rank `rank`'s share of an index file cut into `world` prefix ranges.
 *   bounds == NULL: the entries are cut into `world` runs of about equal BYTE length; bounds_out[world-1] receives the first
 *                   prefix of runs 1..world-1 (2^PREFIX_BITS for an empty tail run): the job's shard bounds.
 *   bounds != NULL: world-1 ascending prefix values; rank r takes the entries with bounds[r-1] <= prefix < bounds[r].
 * The format has no lengths to skip by, so entry starts are RECOGNISED speculatively (a bisection over byte offsets);
 * sequential != 0 walks every entry from the first one instead (always right, reads the file up to the end of the range).
 * The caller must check over ALL ranks that every info.exact is 1 and that the local_entries add up to header_entries,
 * and otherwise repeat the call with sequential = 1 on every rank. K / PREFIX_BITS as for cblx_load. */
int cblx_load_shard_from_file(cblx_ctx* ctx, const char* path, uint32_t rank, uint32_t world, const uint32_t* bounds, int sequential,
                              uint32_t* bounds_out, cblx_shard_info* info);
/* Host-only part of the above (no ctx, no GPU): where `world` prefix ranges cut the entries of an index file. offs[world + 1]
 * = byte offset of every range's first entry ([world] = file size), first[world + 1] = that entry's prefix (2^PREFIX_BITS when
 * nothing follows), *ok = 0 when the speculative search gave up (use sequential = 1). Only k / prefix_bits of params are read. */
int cblx_index_shard_cuts(const cblx_params* params, const char* path, uint32_t world, const uint32_t* bounds, int sequential,
                          uint64_t* offs, uint32_t* first, int* ok);
/* The resident index as a bucket batch: ascending non-empty prefixes, words per prefix, kind (0 = Vec, 1 = Trie) and the
 * suffixes packed to cblx_consts.bytes little-endian bytes each, bucket-major, STORED order inside a bucket. */
typedef struct cblx_bucket_view {
    uint64_t n_buckets, n_words;
    const uint32_t* d_prefix; /* device [n_buckets] */
    const uint32_t* d_count;  /* device [n_buckets] */
    const uint8_t* d_kind;    /* device [n_buckets] */
    const uint8_t* d_suffix;  /* device [n_words * suffix_bytes] */
} cblx_bucket_view;
/* Where nd-1 ascending prefix bounds (host) cut the resident index: destination d = #{i : bounds[i] <= prefix} owns buckets
 * [bucket_split[d], bucket_split[d+1]) and words [word_split[d], word_split[d+1]) of the export (host arrays, nd + 1). */
int cblx_resident_split(cblx_ctx* ctx, const uint32_t* bounds, uint32_t nd, uint64_t* bucket_split, uint64_t* word_split);
/* Writes the whole resident index to caller-owned device arrays of cblx_num_buckets / cblx_count elements (the index stays). */
int cblx_resident_export(cblx_ctx* ctx, uint32_t* d_prefix, uint32_t* d_count, uint8_t* d_kind, uint8_t* d_suffix);
/* Replaces the ctx's contents by the concatenation of n_parts bucket batches (prefixes strictly ascending over the whole
 * concatenation: the pieces of one prefix range received from ranks 0..W-1 in rank order). Kinds and stored order are kept,
 * as WordSet's Deserialize keeps them (src/wordset/mod.rs:398-437). */
int cblx_install_buckets_device(cblx_ctx* ctx, const cblx_bucket_view* parts, uint32_t n_parts);
/* The entries of the serialized index alone — what follows `canonical u8 | varint(n_entries)` in cblx_serialize's bytes — so
 * that several ranks can write one file: size first, then the bytes at `file_off` of an existing file (opened write-only). */
int cblx_serialized_body_size(cblx_ctx* ctx, uint64_t* n_entries, uint64_t* nbytes);
int cblx_write_body_at(cblx_ctx* ctx, const char* path, uint64_t file_off);

/* ---- the multi-GPU build behind this ABI (no reference counterpart; BASELINE.json north_star: prefix space partitioned over the
 * GPUs of a node, one exchange over xGMI after the radix step) -------------------------------------------------------------
 * One process (or thread) per GPU, each with its own cblx_ctx and one cblx_comm. The communicator is RCCL (librccl.so.1 is
 * looked up at run time; rank 0 makes the id with cblx_comm_unique_id and the host program hands it to the other ranks by
 * whatever means it has), or a set of host callbacks that move the bytes (tests; fabrics RCCL does not drive). */
typedef struct cblx_comm cblx_comm;
#define CBLX_COMM_ID_BYTES 128
typedef struct cblx_transport {
    void* user;
    /* in place: vals[i] = sum over ranks of vals[i] (host memory), on every rank */
    int (*all_reduce_sum_u64)(void* user, uint64_t* vals, uint64_t n);
    /* send[d * per_rank ..] goes to rank d, recv[s * per_rank ..] comes from rank s (host memory) */
    int (*all_to_all_u64)(void* user, const uint64_t* send, uint64_t* recv, uint64_t per_rank);
    /* personalised exchange of byte runs in DEVICE memory: bytes [send_off[d], send_off[d+1]) of d_src go to rank d, the bytes
     * from rank s land at [recv_off[s], recv_off[s+1]) of d_dst (world + 1 host offsets each; the rank's own run included).
     * Complete on return. */
    int (*exchange)(void* user, const uint8_t* d_src, const uint64_t* send_off, uint8_t* d_dst, const uint64_t* recv_off);
} cblx_transport;
int cblx_comm_unique_id(uint8_t id[CBLX_COMM_ID_BYTES]);
int cblx_comm_init_rccl(cblx_comm** out, const uint8_t* id, uint32_t rank, uint32_t world, int32_t device);
int cblx_comm_init_transport(cblx_comm** out, const cblx_transport* t, uint32_t rank, uint32_t world, int32_t device);
void cblx_comm_destroy(cblx_comm* comm);
const char* cblx_comm_last_error(const cblx_comm* comm);
typedef struct cblx_exchange_stats { uint64_t sent_bytes, recv_bytes, messages; } cblx_exchange_stats; /* own runs excluded */
int cblx_comm_stats(cblx_comm* comm, cblx_exchange_stats* out, int reset);
/* What crosses the links in cblx_sharded_insert_seqs_device (every rank of a job sets the same):
 *   CBLX_PROTO_BINS: the exchange sits between the first and the second partition pass. The sender runs KRN-1 and
 *     the first pass on bins that refine that pass's digit by the destination rank; 8-byte records (16 for words that keep
 *     a 64-bit hi part) + 1 digit byte per word cross the links, the receiver runs the remaining passes and the bucket
 *     kernels on what arrived. No pass is added to the one-GPU pipeline and nothing is copied: the choice when the kernels
 *     are the bound (8 GPUs). Needs PREFIX_BITS >= 9, else SORTED is used.
 *   CBLX_PROTO_SORTED: the sender partitions completely; non-empty prefixes, counts and the suffixes packed to
 *     cblx_consts.bytes cross the links (6.1 B per word at K = 31 / PREFIX_BITS = 24), the receiver merges the batches run
 *     by run (one more pass over the words): the choice when the links are the bound (2-4 GPUs).
 *   CBLX_PROTO_AUTO (what a new communicator is set to): SORTED on 2 - 4 ranks, BINS otherwise. Between 2 - 4 GPUs every pair shares ONE link
 *     and the bytes on it bound the job; rehearsed against a paced wire (profiles/r05_wire_emulated.md, cfg 3, 55 GB/s per link): 2 ranks
 *     101 ms SORTED / 127 ms BINS, 4 ranks 69 / 71 ms (the two meet near 60 GB/s per link), 8 ranks BINS 47 ms. cblx_comm_protocol_used: what the last sharded insert ran on. */
#define CBLX_PROTO_SORTED 0u
#define CBLX_PROTO_BINS 1u
#define CBLX_PROTO_AUTO 2u
/*   CBLX_PROTO_REPLICATE (round 6): READS cross the links, not words — every rank packs its reads into bit planes on the device (3 bits per
 *     base = 0.3 bytes per k-mer), planes and offsets are all-gathered, and every rank runs KRN-1 and the first partition pass over ALL ranks'
 *     reads, keeping the words of its own prefix range; the receiver steps are those of BINS (groups, FINE bins). W encodes per rank instead
 *     of one and next to nothing on the wire: the choice where one link per pair of GPUs bounds the other protocols (2 - 3 GPUs). Falls back to
 *     BINS together with every other rank on a non-empty index or bounds the bins refuse. */
#define CBLX_PROTO_REPLICATE 3u
int cblx_comm_set_protocol(cblx_comm* comm, uint32_t protocol);
int cblx_comm_protocol_used(const cblx_comm* comm, uint32_t* out);
/* The receiver of CBLX_PROTO_BINS in GROUPS (no reference counterpart; same result, another schedule): every rank's prefix range is
 * cut into `groups` parts of about equal sampled mass, the senders' first pass also separates the groups, the data crosses the links
 * group-major, and the receiver runs the remaining passes + bucket kernels of group g while groups g+1.. are still on the wire.
 * 0 = default (CBLX_RECV_GROUPS in the environment, else 4), 1 = off (everything waits for the last record), at most 14. Every rank
 * of a job sets the same. cblx_comm_groups_used: groups the last cblx_sharded_insert_seqs_device of this rank worked through
 * (0: it took the ungrouped path — index not empty, ranges too narrow for the cuts, one rank). */
/* Rehearsal of ONE rank of a `world`-GPU job on one GPU (dev / bench tooling, no reference counterpart: tools/emulate_wire.py). Ranks
 * 1 .. world-1 are run one after the other on communicators that RECORD, in the process-wide store `store_id`, what each would send
 * to rank 0; a rank-0 communicator then REPLAYS them: its collectives are answered from the records and the bytes of every exchange
 * are copied into its receive arena on a side stream held back until a wire of `link_gbps` GB/s per source rank would have delivered
 * them (0 = as fast as the copies go). All ranks of a rehearsal hold equally many reads and make the same calls with the same bounds
 * and group setting. */
int cblx_comm_init_sim(cblx_comm** out, uint32_t rank, uint32_t world, int32_t device, uint64_t store_id, double link_gbps);
int cblx_sim_store_free(uint64_t store_id);
int cblx_comm_set_recv_groups(cblx_comm* comm, uint32_t groups);
int cblx_comm_groups_used(const cblx_comm* comm, uint32_t* out);
/* ... and how many of those groups sorted 16 prefix bits behind the senders' first pass instead of PREFIX_BITS - 8 (PREFIX_BITS > 24: the first
 * pass runs on FINE bins that cut a narrow group's prefix range into aligned blocks of 2^16 prefixes, so its receiver needs two partition
 * passes instead of three; CBLX_FINE_BINS=0 in the environment switches that off). Same result either way. */
int cblx_comm_groups_fine(const cblx_comm* comm, uint32_t* out);
/* CBL::insert_seq for every sequence of THIS rank's shard, into an index sharded by prefix range over the ranks of `comm`
 * (every rank makes the same call with its own shard). The shard is consumed in n_slices slices, reads
 * [slice_cuts[s], slice_cuts[s+1]) (n_slices + 1 ascending values, the same NUMBER of slices on every rank): the exchange of
 * a slice overlaps the kernels of the next one, and the job's stream order is slice-major, rank-minor — the order of the file
 * when its blocks were dealt to the ranks cyclically (cblx_stage_fastx_blocks). bounds[world - 1]: the shard bounds (rank r
 * owns bounds[r-1] <= prefix < bounds[r]); *bounds_valid = 0 on the first call lets the ranks choose them together (quantiles
 * of a sampled prefix histogram: necklace prefixes are heavily skewed) and sets it to 1. Afterwards the ctx of rank r holds
 * range r of the index: count / serialize / cblx_write_body_at work per range. */
int cblx_sharded_insert_seqs_device(cblx_ctx* ctx, cblx_comm* comm, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n,
                                    const uint64_t* slice_cuts, uint32_t n_slices, uint32_t* bounds, int* bounds_valid);

/* CBL::count / is_empty / is_canonical (src/cbl.rs:164-177). */
int cblx_count(cblx_ctx* ctx, uint64_t* out);
int cblx_num_buckets(cblx_ctx* ctx, uint64_t* out); /* tiered.len() = number of non-empty prefixes */
int cblx_is_empty(cblx_ctx* ctx, int* out);
int cblx_is_canonical(const cblx_ctx* ctx, int* out);

/* Serialize (derive on CBL src/cbl.rs:40-54 + WordSet::serialize src/wordset/mod.rs:382-396) with
 * bincode DefaultOptions + varint (src/cbl.rs:132-135): the exact bytes of CBL::save_to_file. */
int cblx_serialized_size(cblx_ctx* ctx, uint64_t* nbytes);
int cblx_serialize(cblx_ctx* ctx, uint8_t* buf, uint64_t cap, uint64_t* written);
int cblx_save_to_file(cblx_ctx* ctx, const char* path);
/* Deserialize (src/wordset/mod.rs:398-437; reject_trailing_bytes): replaces the ctx's contents.
 * K / PREFIX_BITS are not stored in the file (compile-time on both sides in the reference). */
int cblx_load(cblx_ctx* ctx, const uint8_t* data, uint64_t len);
int cblx_load_from_file(cblx_ctx* ctx, const char* path);

/* `self |= other` (src/cbl.rs:433-449 -> src/wordset/set_ops.rs:123-157). */
int cblx_merge_assign(cblx_ctx* self, cblx_ctx* other);
/* dst = what `self |= other` would leave in self, with self itself untouched (and other exactly as `|=` leaves it: its Vec buckets that
 * meet a bucket of self sorted). No reference counterpart as a call — it is `dst = self.clone(); dst |= other`
 * (/root/reference/src/cbl.rs:433-449) without the copy: the device merge writes a new arena anyway. dst's previous content is dropped;
 * dst, self and other are three different contexts of equal K / PREFIX_BITS / canonical, dst and self on the same device. */
int cblx_merge_from(cblx_ctx* dst, cblx_ctx* self, cblx_ctx* other);

/* Walk the resident buckets in ascending prefix order (lets a Rust shim rebuild a WordSet, and tests
 * compare bucket contents): kind 0 = Vec (stored = first-occurrence order), 1 = Trie (ascending).
 * Suffix i of the bucket is (hi[i] << 64) | lo[i]; hi is NULL when SUFFIX_BITS <= 64. Return non-zero
 * from the callback to stop. */
typedef int (*cblx_bucket_cb)(void* user, uint32_t prefix, int kind, uint64_t len, const uint64_t* lo,
                              const uint64_t* hi);
int cblx_export_buckets(cblx_ctx* ctx, cblx_bucket_cb cb, void* user);

/* CBL::contains_seq (src/cbl.rs:311-324): one byte (0/1) per k-mer of one sequence, into `out[cap]`. */
int cblx_contains_seq(cblx_ctx* ctx, const uint8_t* seq, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* n);
/* The same for a batch of sequences in one pass (the loop of `cbl query`, examples/cbl.rs:205-228): flags of sequence 0,
 * then sequence 1, ... (`out` may be NULL when only the tallies are wanted); *n_out = k-mers queried, *positive = flags
 * set. Every sequence must hold at least K bytes (CBLX_ESHORT otherwise, nothing is queried).
 * _device: d_bases (16-byte aligned, 16 readable bytes past the end) / d_offsets / d_out are device pointers. */
int cblx_contains_seqs(cblx_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n, uint8_t* out, uint64_t cap,
                       uint64_t* n_out, uint64_t* positive);
int cblx_contains_seqs_device(cblx_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, uint8_t* d_out,
                              uint64_t cap, uint64_t* n_out, uint64_t* positive);
/* `cbl query <index> <fastx>` (examples/cbl.rs:205-228): contains_seq for every record of a FASTA/FASTQ(.gz) file, read like
 * cblx_insert_fastx_file; *total = k-mers queried, *positive = those found. The index is not modified. */
int cblx_query_fastx_file(cblx_ctx* ctx, const char* path, uint64_t* n_records, uint64_t* total, uint64_t* positive);
/* CBL::contains_all (src/cbl.rs:293-307): *out = 1 iff every k-mer of the sequence is in the set. */
int cblx_contains_all(cblx_ctx* ctx, const uint8_t* seq, uint64_t len, int* out);

/* Packed k-mers: k-mer i is IntKmer::to_int() (2K bits, first base most significant, src/kmer.rs:200-202) given as
 * lo[i] (low 64 bits) and hi[i] (the rest; `hi` may be NULL when K <= 31, must not be when K >= 33).
 *   cblx_insert_kmers   = n successive CBL::insert calls (src/cbl.rs:226-228 -> get_word :199-206 -> WordSet::insert
 *                         src/wordset/mod.rs:97-120); was_absent[i] (may be NULL) is the i-th call's return value:
 *                         1 iff the k-mer was neither in the set nor earlier in this batch.
 *   cblx_contains_kmers = CBL::contains (src/cbl.rs:219-221) for each k-mer, one byte (0/1) each.
 * In a canonical index the k-mer is replaced by its canonical form first, as in the reference. A k-mer with bits set
 * above 2K is refused (CBLX_EINVAL) and nothing is inserted. */
int cblx_insert_kmers(cblx_ctx* ctx, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* was_absent);
int cblx_contains_kmers(cblx_ctx* ctx, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out);
/* CBL::iter (src/cbl.rs:358-361): every k-mer of the set, packed as above, in the reference's iteration order
 * (prefixes ascending; a Vec bucket in stored order, a Trie bucket ascending), recovered from its word by
 * revert_necklace_pos (src/necklace/mod.rs:29-31). *n = count(); `hi` may be NULL when K <= 31. */
int cblx_export_kmers(cblx_ctx* ctx, uint64_t* lo, uint64_t* hi, uint64_t cap, uint64_t* n);
/* Bucket table only (CBL::buckets_sizes src/cbl.rs:370-373 and the statistics built on it): per non-empty prefix in
 * ascending order its value, the bucket's length and kind (0 = Vec, 1 = Trie). Any of the arrays may be NULL. */
int cblx_bucket_sizes(cblx_ctx* ctx, uint32_t* prefix, uint32_t* len, uint8_t* kind, uint64_t cap, uint64_t* n);

/* Full-size parity properties (no reference counterpart): an order-independent checksum of the set (sum over the
 * resident words of a 64-bit hash, mod 2^64), the same sum over a device array of words, and a structural check of the
 * resident buckets (Trie buckets strictly ascending, Vec buckets pairwise distinct, and with `strict` the insert-only
 * rule Vec <= 1024 < Trie of src/wordset/mod.rs:240-244). */
int cblx_checksum(cblx_ctx* ctx, uint64_t* sum);
int cblx_checksum_words_device(cblx_ctx* ctx, const uint64_t* d_lo, const void* d_hi, uint64_t n, uint64_t* sum);
int cblx_validate(cblx_ctx* ctx, int strict, uint64_t* violations);

/* Derived constants (src/cbl.rs:16-32,65-67), for shims and tests. */
typedef struct cblx_consts {
    uint32_t kmer_bits, pos_bits, word_bits, suffix_bits, bytes, chunk_size, threshold, hi_bytes;
} cblx_consts;
int cblx_get_consts(const cblx_ctx* ctx, cblx_consts* out);

/* Per-stage device time (ms, HIP events on the ctx's stream) accumulated since the last reset; needs
 * CBLX_FLAG_PROFILE. names[i] points to static storage. Returns the number of stages in *n (<= cap). */
int cblx_stage_times(cblx_ctx* ctx, const char** names, double* ms, uint64_t* launches, uint32_t cap, uint32_t* n);
int cblx_stage_times_reset(cblx_ctx* ctx);
/* Words the kernels of every stage were given since the last reset, in the order of cblx_stage_times, where the pipeline counts them
 * (today: radix_scatter — records through a partition pass, summed over the passes: how many a record takes depends on the route
 * (PREFIX_BITS > 24: the FINE-bins build sorts most of them in three passes, the rest in four); and, with CBLX_FLAG_PROFILE, `self |= other`,
 * whose bucket classes split the words between the stages — merge_gather: the words of the one-sided buckets it copies, bucket_medium:
 * the words of both-sided buckets with a Vec side, bucket_big: the OUTPUT words of the Trie |= Trie unions, bucket_huge: the rest).
 * 0 = not counted: the caller prices the stage on the whole batch. */
int cblx_stage_units(cblx_ctx* ctx, uint64_t* units, uint32_t cap, uint32_t* n);
/* k-mers (words) consumed by insert calls since creation — the numerator of the throughput metric. */
int cblx_kmers_inserted(cblx_ctx* ctx, uint64_t* out);
/* Batches this ctx built through the FINE-bins route (PREFIX_BITS > 24, empty index, a batch of CBLX_FINE_MIN k-mers or more — default
 * 2^22 —, CBLX_FINE_BINS != 0): the first partition pass runs on 253 intervals of the prefix space — aligned blocks of 2^16 prefixes
 * where the necklace prefixes are dense — so that most words need two more passes instead of three. Same index either way. */
int cblx_fine_builds(cblx_ctx* ctx, uint64_t* out);
/* Release cached device workspace (kept between flushes to avoid hipMalloc in the hot path). */
int cblx_trim(cblx_ctx* ctx);
/* Back to the state of CBL::new(): drops the resident index and anything enqueued, keeps the cached workspace. */
int cblx_clear(cblx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* CBLX_H */
