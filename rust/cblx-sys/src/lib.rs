//! cblx-sys — raw `extern "C"` bindings to `libcblx.so`, one item per declaration of `include/cblx.h` (same order).
//!
//! NOT compiled in the image this repository was built in (no Rust toolchain there): shipped as source, and kept in step
//! with the header by `tests/test_abi_and_host_units.py::test_rust_sys_crate_declares_every_symbol`, which diffs the
//! function names below against the header's. What each function does, and the reference code it stands for
//! (`/root/reference/src/cbl.rs` etc.), is documented in the header; it is not repeated here.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const CBLX_ABI_VERSION: u32 = 3;

pub const CBLX_OK: c_int = 0;
pub const CBLX_EINVAL: c_int = 1;
pub const CBLX_ESHORT: c_int = 2;
pub const CBLX_EFORMAT: c_int = 3;
pub const CBLX_EDEVICE: c_int = 4;
pub const CBLX_ENOMEM: c_int = 5;
pub const CBLX_ERANGE: c_int = 6;

pub const CBLX_FLAG_PROFILE: u32 = 1;
pub const CBLX_COMM_ID_BYTES: usize = 128;
pub const CBLX_PROTO_SORTED: u32 = 0;
pub const CBLX_PROTO_BINS: u32 = 1;
pub const CBLX_PROTO_AUTO: u32 = 2;
pub const CBLX_PROTO_REPLICATE: u32 = 3;

#[repr(C)]
pub struct cblx_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cblx_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cblx_params {
    pub k: u32,
    pub prefix_bits: u32,
    pub canonical: u32,
    pub device: i32,
    pub flags: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cblx_batch_view {
    pub n_buckets: u64,
    pub n_words: u64,
    pub d_prefix: *const u32,
    pub d_count: *const u32,
    pub d_suffix: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cblx_shard_info {
    pub header_entries: u64,
    pub local_entries: u64,
    pub begin_off: u64,
    pub end_off: u64,
    pub first_prefix: u32,
    pub last_prefix: u32,
    pub exact: u32,
    pub canonical: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cblx_bucket_view {
    pub n_buckets: u64,
    pub n_words: u64,
    pub d_prefix: *const u32,
    pub d_count: *const u32,
    pub d_kind: *const u8,
    pub d_suffix: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct cblx_transport {
    pub user: *mut c_void,
    pub all_reduce_sum_u64: Option<unsafe extern "C" fn(user: *mut c_void, vals: *mut u64, n: u64) -> c_int>,
    pub all_to_all_u64: Option<unsafe extern "C" fn(user: *mut c_void, send: *const u64, recv: *mut u64, per_rank: u64) -> c_int>,
    pub exchange: Option<
        unsafe extern "C" fn(user: *mut c_void, d_src: *const u8, send_off: *const u64, d_dst: *mut u8, recv_off: *const u64) -> c_int,
    >,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cblx_exchange_stats {
    pub sent_bytes: u64,
    pub recv_bytes: u64,
    pub messages: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cblx_consts {
    pub kmer_bits: u32,
    pub pos_bits: u32,
    pub word_bits: u32,
    pub suffix_bits: u32,
    pub bytes: u32,
    pub chunk_size: u32,
    pub threshold: u32,
    pub hi_bytes: u32,
}

pub type cblx_bucket_cb =
    Option<unsafe extern "C" fn(user: *mut c_void, prefix: u32, kind: c_int, len: u64, lo: *const u64, hi: *const u64) -> c_int>;

extern "C" {
    pub fn cblx_abi_version() -> u32;
    pub fn cblx_last_global_error() -> *const c_char;

    pub fn cblx_create(params: *const cblx_params, out: *mut *mut cblx_ctx) -> c_int;
    pub fn cblx_destroy(ctx: *mut cblx_ctx);
    pub fn cblx_last_error(ctx: *const cblx_ctx) -> *const c_char;

    pub fn cblx_insert_seq(ctx: *mut cblx_ctx, seq: *const u8, len: u64) -> c_int;
    pub fn cblx_insert_seqs(ctx: *mut cblx_ctx, bases: *const u8, offsets: *const u64, n: u64) -> c_int;
    pub fn cblx_insert_seqs_device(ctx: *mut cblx_ctx, d_bases: *const u8, d_offsets: *const u64, n: u64) -> c_int;
    pub fn cblx_insert_fastx_file(ctx: *mut cblx_ctx, path: *const c_char, n_records: *mut u64) -> c_int;
    pub fn cblx_flush(ctx: *mut cblx_ctx) -> c_int;

    pub fn cblx_stage_fastx_blocks(
        ctx: *mut cblx_ctx,
        path: *const c_char,
        block: u64,
        rank: u32,
        world: u32,
        d_bases: *mut *const u8,
        d_offsets: *mut *const u64,
        n_staged: *mut u64,
        n_in_file: *mut u64,
    ) -> c_int;
    pub fn cblx_stage_release(ctx: *mut cblx_ctx) -> c_int;
    pub fn cblx_stage_fastx_blocks_comm(
        ctx: *mut cblx_ctx,
        comm: *mut cblx_comm,
        path: *const c_char,
        block: *mut u64,
        slices: u32,
        d_bases: *mut *const u8,
        d_offsets: *mut *const u64,
        n_staged: *mut u64,
        n_in_file: *mut u64,
    ) -> c_int;

    pub fn cblx_insert_words_device(ctx: *mut cblx_ctx, d_lo: *const u64, d_hi: *const c_void, n: u64) -> c_int;
    pub fn cblx_seq_words_device(
        ctx: *mut cblx_ctx,
        d_bases: *const u8,
        d_offsets: *const u64,
        n: u64,
        d_lo: *mut u64,
        d_hi: *mut c_void,
        cap: u64,
        n_words: *mut u64,
    ) -> c_int;
    pub fn cblx_partition_words_device(
        ctx: *mut cblx_ctx,
        d_lo: *const u64,
        d_hi: *const c_void,
        n: u64,
        bounds: *const u32,
        nd: u32,
        d_out_lo: *mut u64,
        d_out_hi: *mut c_void,
        counts: *mut u64,
    ) -> c_int;
    pub fn cblx_seq_words_partitioned_device(
        ctx: *mut cblx_ctx,
        d_bases: *const u8,
        d_offsets: *const u64,
        n: u64,
        bounds: *const u32,
        nd: u32,
        d_out_lo: *mut u64,
        d_out_hi: *mut c_void,
        cap: u64,
        counts: *mut u64,
        n_words: *mut u64,
    ) -> c_int;

    pub fn cblx_sorted_batch_begin(
        ctx: *mut cblx_ctx,
        d_bases: *const u8,
        d_offsets: *const u64,
        n: u64,
        bounds: *const u32,
        nd: u32,
        bucket_split: *mut u64,
        word_split: *mut u64,
    ) -> c_int;
    pub fn cblx_sorted_batch_export(ctx: *mut cblx_ctx, d_prefix: *mut u32, d_count: *mut u32, d_suffix: *mut u8) -> c_int;
    pub fn cblx_insert_sorted_batches_device(ctx: *mut cblx_ctx, batches: *const cblx_batch_view, n_batches: u32) -> c_int;

    pub fn cblx_load_shard_from_file(
        ctx: *mut cblx_ctx,
        path: *const c_char,
        rank: u32,
        world: u32,
        bounds: *const u32,
        sequential: c_int,
        bounds_out: *mut u32,
        info: *mut cblx_shard_info,
    ) -> c_int;
    pub fn cblx_index_shard_cuts(
        params: *const cblx_params,
        path: *const c_char,
        world: u32,
        bounds: *const u32,
        sequential: c_int,
        offs: *mut u64,
        first: *mut u32,
        ok: *mut c_int,
    ) -> c_int;
    pub fn cblx_resident_split(ctx: *mut cblx_ctx, bounds: *const u32, nd: u32, bucket_split: *mut u64, word_split: *mut u64) -> c_int;
    pub fn cblx_resident_export(ctx: *mut cblx_ctx, d_prefix: *mut u32, d_count: *mut u32, d_kind: *mut u8, d_suffix: *mut u8) -> c_int;
    pub fn cblx_install_buckets_device(ctx: *mut cblx_ctx, parts: *const cblx_bucket_view, n_parts: u32) -> c_int;
    pub fn cblx_serialized_body_size(ctx: *mut cblx_ctx, n_entries: *mut u64, nbytes: *mut u64) -> c_int;
    pub fn cblx_write_body_at(ctx: *mut cblx_ctx, path: *const c_char, file_off: u64) -> c_int;

    pub fn cblx_comm_unique_id(id: *mut u8) -> c_int;
    pub fn cblx_comm_init_rccl(out: *mut *mut cblx_comm, id: *const u8, rank: u32, world: u32, device: i32) -> c_int;
    pub fn cblx_comm_init_transport(out: *mut *mut cblx_comm, t: *const cblx_transport, rank: u32, world: u32, device: i32) -> c_int;
    pub fn cblx_comm_destroy(comm: *mut cblx_comm);
    pub fn cblx_comm_last_error(comm: *const cblx_comm) -> *const c_char;
    pub fn cblx_comm_stats(comm: *mut cblx_comm, out: *mut cblx_exchange_stats, reset: c_int) -> c_int;
    pub fn cblx_comm_set_protocol(comm: *mut cblx_comm, protocol: u32) -> c_int;
    pub fn cblx_comm_init_sim(out: *mut *mut cblx_comm, rank: u32, world: u32, device: i32, store_id: u64, link_gbps: f64) -> c_int;
    pub fn cblx_sim_store_free(store_id: u64) -> c_int;
    pub fn cblx_comm_set_recv_groups(comm: *mut cblx_comm, groups: u32) -> c_int;
    pub fn cblx_comm_groups_used(comm: *const cblx_comm, out: *mut u32) -> c_int;
    pub fn cblx_comm_groups_fine(comm: *const cblx_comm, out: *mut u32) -> c_int;
    pub fn cblx_comm_protocol_used(comm: *const cblx_comm, out: *mut u32) -> c_int;
    pub fn cblx_fine_builds(ctx: *mut cblx_ctx, out: *mut u64) -> c_int;
    pub fn cblx_sharded_insert_seqs_device(
        ctx: *mut cblx_ctx,
        comm: *mut cblx_comm,
        d_bases: *const u8,
        d_offsets: *const u64,
        n: u64,
        slice_cuts: *const u64,
        n_slices: u32,
        bounds: *mut u32,
        bounds_valid: *mut c_int,
    ) -> c_int;

    pub fn cblx_count(ctx: *mut cblx_ctx, out: *mut u64) -> c_int;
    pub fn cblx_num_buckets(ctx: *mut cblx_ctx, out: *mut u64) -> c_int;
    pub fn cblx_is_empty(ctx: *mut cblx_ctx, out: *mut c_int) -> c_int;
    pub fn cblx_is_canonical(ctx: *const cblx_ctx, out: *mut c_int) -> c_int;

    pub fn cblx_serialized_size(ctx: *mut cblx_ctx, nbytes: *mut u64) -> c_int;
    pub fn cblx_serialize(ctx: *mut cblx_ctx, buf: *mut u8, cap: u64, written: *mut u64) -> c_int;
    pub fn cblx_save_to_file(ctx: *mut cblx_ctx, path: *const c_char) -> c_int;
    pub fn cblx_load(ctx: *mut cblx_ctx, data: *const u8, len: u64) -> c_int;
    pub fn cblx_load_from_file(ctx: *mut cblx_ctx, path: *const c_char) -> c_int;

    pub fn cblx_merge_assign(this: *mut cblx_ctx, other: *mut cblx_ctx) -> c_int;
    pub fn cblx_merge_from(dst: *mut cblx_ctx, this: *mut cblx_ctx, other: *mut cblx_ctx) -> c_int;
    pub fn cblx_stage_units(ctx: *mut cblx_ctx, units: *mut u64, cap: u32, n: *mut u32) -> c_int;

    pub fn cblx_export_buckets(ctx: *mut cblx_ctx, cb: cblx_bucket_cb, user: *mut c_void) -> c_int;

    pub fn cblx_contains_seq(ctx: *mut cblx_ctx, seq: *const u8, len: u64, out: *mut u8, cap: u64, n: *mut u64) -> c_int;
    pub fn cblx_contains_seqs(
        ctx: *mut cblx_ctx,
        bases: *const u8,
        offsets: *const u64,
        n: u64,
        out: *mut u8,
        cap: u64,
        n_out: *mut u64,
        positive: *mut u64,
    ) -> c_int;
    pub fn cblx_contains_seqs_device(
        ctx: *mut cblx_ctx,
        d_bases: *const u8,
        d_offsets: *const u64,
        n: u64,
        d_out: *mut u8,
        cap: u64,
        n_out: *mut u64,
        positive: *mut u64,
    ) -> c_int;
    pub fn cblx_query_fastx_file(ctx: *mut cblx_ctx, path: *const c_char, n_records: *mut u64, total: *mut u64, positive: *mut u64) -> c_int;
    pub fn cblx_contains_all(ctx: *mut cblx_ctx, seq: *const u8, len: u64, out: *mut c_int) -> c_int;

    pub fn cblx_insert_kmers(ctx: *mut cblx_ctx, lo: *const u64, hi: *const u64, n: u64, was_absent: *mut u8) -> c_int;
    pub fn cblx_contains_kmers(ctx: *mut cblx_ctx, lo: *const u64, hi: *const u64, n: u64, out: *mut u8) -> c_int;
    pub fn cblx_export_kmers(ctx: *mut cblx_ctx, lo: *mut u64, hi: *mut u64, cap: u64, n: *mut u64) -> c_int;
    pub fn cblx_bucket_sizes(ctx: *mut cblx_ctx, prefix: *mut u32, len: *mut u32, kind: *mut u8, cap: u64, n: *mut u64) -> c_int;

    pub fn cblx_checksum(ctx: *mut cblx_ctx, sum: *mut u64) -> c_int;
    pub fn cblx_checksum_words_device(ctx: *mut cblx_ctx, d_lo: *const u64, d_hi: *const c_void, n: u64, sum: *mut u64) -> c_int;
    pub fn cblx_validate(ctx: *mut cblx_ctx, strict: c_int, violations: *mut u64) -> c_int;

    pub fn cblx_get_consts(ctx: *const cblx_ctx, out: *mut cblx_consts) -> c_int;

    pub fn cblx_stage_times(ctx: *mut cblx_ctx, names: *mut *const c_char, ms: *mut f64, launches: *mut u64, cap: u32, n: *mut u32) -> c_int;
    pub fn cblx_stage_times_reset(ctx: *mut cblx_ctx) -> c_int;
    pub fn cblx_kmers_inserted(ctx: *mut cblx_ctx, out: *mut u64) -> c_int;
    pub fn cblx_trim(ctx: *mut cblx_ctx) -> c_int;
    pub fn cblx_clear(ctx: *mut cblx_ctx) -> c_int;
}
