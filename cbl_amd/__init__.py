"""cbl_amd — MI355X-native bulk k-mer insertion for CBL indexes (host-side mirror of the reference API).

`CBL` mirrors the method surface of the reference's `CBL<K, T, PREFIX_BITS>` for the build / insert / merge path
(/root/reference/src/cbl.rs:71-79 new/new_canonical, :127-160 save_to_file/load_from_file, :164-177
count/is_empty/is_canonical, :311-324 contains_seq, :328-339 insert_seq, :433-449 `|=`) on top of the C ABI in
include/cblx.h (libcblx.so, hand-written HIP for gfx950). There is NO CPU fallback: importing works anywhere, but
creating a `CBL` without the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["CBLX_LIB_PATH"]) if os.environ.get("CBLX_LIB_PATH") else _HERE / "libcblx.so"  # override: tuning variants (tools/)
_LIB = None

OK, EINVAL, ESHORT, EFORMAT, EDEVICE, ENOMEM, ERANGE = range(7)
FLAG_PROFILE = 1


class CblxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("prefix_bits", C.c_uint32), ("canonical", C.c_uint32), ("device", C.c_int32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Consts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kmer_bits", "pos_bits", "word_bits", "suffix_bits", "bytes", "chunk_size",
                                          "threshold", "hi_bytes")]


class ShardInfo(C.Structure):
    _fields_ = [("header_entries", C.c_uint64), ("local_entries", C.c_uint64), ("begin_off", C.c_uint64), ("end_off", C.c_uint64),
                ("first_prefix", C.c_uint32), ("last_prefix", C.c_uint32), ("exact", C.c_uint32), ("canonical", C.c_uint32)]


class BucketView(C.Structure):
    _fields_ = [("n_buckets", C.c_uint64), ("n_words", C.c_uint64), ("d_prefix", C.c_void_p), ("d_count", C.c_void_p), ("d_kind", C.c_void_p),
                ("d_suffix", C.c_void_p)]


class ExchangeStats(C.Structure):
    _fields_ = [("sent_bytes", C.c_uint64), ("recv_bytes", C.c_uint64), ("messages", C.c_uint64)]


_ALL_REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)
_ALL_TO_ALL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64)
_EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_reduce_sum_u64", _ALL_REDUCE_CB), ("all_to_all_u64", _ALL_TO_ALL_CB), ("exchange", _EXCHANGE_CB)]


class BatchView(C.Structure):
    _fields_ = [("n_buckets", C.c_uint64), ("n_words", C.c_uint64), ("d_prefix", C.c_void_p), ("d_count", C.c_void_p), ("d_suffix", C.c_void_p)]


BUCKET_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))

ABI_VERSION = 3  # include/cblx.h CBLX_ABI_VERSION

# name -> (restype, argtypes): every symbol include/cblx.h declares
SIGNATURES = {
    "cblx_abi_version": (C.c_uint32, []),
    "cblx_last_global_error": (C.c_char_p, []),
    "cblx_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "cblx_destroy": (None, [C.c_void_p]),
    "cblx_last_error": (C.c_char_p, [C.c_void_p]),
    "cblx_insert_seq": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "cblx_insert_seqs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "cblx_insert_seqs_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "cblx_insert_fastx_file": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]),
    "cblx_stage_fastx_blocks": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cblx_stage_fastx_blocks_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cblx_stage_release": (C.c_int, [C.c_void_p]),
    "cblx_flush": (C.c_int, [C.c_void_p]),
    "cblx_insert_words_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "cblx_seq_words_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64)]),
    "cblx_partition_words_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "cblx_seq_words_partitioned_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                                    C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_sorted_batch_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]),
    "cblx_sorted_batch_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cblx_insert_sorted_batches_device": (C.c_int, [C.c_void_p, C.POINTER(BatchView), C.c_uint32]),
    "cblx_load_shard_from_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(ShardInfo)]),
    "cblx_index_shard_cuts": (C.c_int, [C.POINTER(Params), C.c_char_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "cblx_resident_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cblx_resident_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cblx_install_buckets_device": (C.c_int, [C.c_void_p, C.POINTER(BucketView), C.c_uint32]),
    "cblx_serialized_body_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cblx_write_body_at": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "cblx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cblx_comm_init_rccl": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32]),
    "cblx_comm_init_transport": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Transport), C.c_uint32, C.c_uint32, C.c_int32]),
    "cblx_comm_destroy": (None, [C.c_void_p]),
    "cblx_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "cblx_comm_stats": (C.c_int, [C.c_void_p, C.POINTER(ExchangeStats), C.c_int]),
    "cblx_comm_set_protocol": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cblx_comm_init_sim": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, C.c_double]),
    "cblx_sim_store_free": (C.c_int, [C.c_uint64]),
    "cblx_comm_set_recv_groups": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cblx_comm_groups_used": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "cblx_comm_groups_fine": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "cblx_comm_protocol_used": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "cblx_fine_builds": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_sharded_insert_seqs_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                                  C.POINTER(C.c_int)]),
    "cblx_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_num_buckets": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_is_empty": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cblx_is_canonical": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cblx_serialized_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_serialize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cblx_save_to_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cblx_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "cblx_load_from_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cblx_merge_assign": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cblx_merge_from": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cblx_stage_units": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]),
    "cblx_export_buckets": (C.c_int, [C.c_void_p, BUCKET_CB, C.c_void_p]),
    "cblx_contains_seq": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cblx_contains_seqs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64)]),
    "cblx_contains_seqs_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]),
    "cblx_query_fastx_file": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cblx_contains_all": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_int)]),
    "cblx_insert_kmers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cblx_contains_kmers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cblx_export_kmers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cblx_bucket_sizes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cblx_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_checksum_words_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cblx_validate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "cblx_get_consts": (C.c_int, [C.c_void_p, C.POINTER(Consts)]),
    "cblx_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_uint32,
                                   C.POINTER(C.c_uint32)]),
    "cblx_stage_times_reset": (C.c_int, [C.c_void_p]),
    "cblx_kmers_inserted": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cblx_trim": (C.c_int, [C.c_void_p]),
    "cblx_clear": (C.c_int, [C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load libcblx.so (built in-tree by __graft_entry__.build()). Fails loudly when it is missing."""
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). cbl_amd has no CPU fallback.")
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("CBLX_LIB_PATH") and not hasattr(L, name):
                continue  # an older build loaded for comparison (tools/): its missing entry points are simply not bound
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if not os.environ.get("CBLX_LIB_PATH") and L.cblx_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports ABI version {L.cblx_abi_version()}, this binding was written against {ABI_VERSION} "
                              "(include/cblx.h CBLX_ABI_VERSION): rebuild with __graft_entry__.build()")
        _LIB = L
    return _LIB


def _ptr(x):
    """Raw address of a torch tensor / numpy array / int."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    return int(x)


def index_shard_cuts(path, k: int, prefix_bits: int, world: int, bounds=None, sequential: bool = False):
    """Where `world` prefix ranges cut the entries of an index file (host only, no GPU): (byte offsets[world + 1],
    first prefixes[world + 1], ok)."""
    import numpy as np

    L = lib()
    p = Params(k, prefix_bits, 0, -1, 0, 0)
    b = np.ascontiguousarray(bounds, dtype=np.uint32) if bounds is not None else None
    offs = np.zeros(world + 1, dtype=np.uint64)
    first = np.zeros(world + 1, dtype=np.uint32)
    ok = C.c_int(0)
    rc = L.cblx_index_shard_cuts(C.byref(p), os.fsencode(path), world, b.ctypes.data if b is not None and len(b) else None, int(sequential),
                                 offs.ctypes.data, first.ctypes.data, C.byref(ok))
    if rc != OK:
        raise CblxError(rc, L.cblx_last_global_error().decode())
    return offs, first, bool(ok.value)


class Comm:
    """One rank's communicator of the multi-GPU build behind the C ABI (include/cblx.h: cblx_comm).

    `Comm.rccl(id, rank, world, device)`: RCCL over xGMI, `id` = Comm.unique_id() of rank 0 shipped to every rank by the host
    program. `Comm.over_group(dist, rank, world, device)`: host callbacks that move the bytes through a torch.distributed-like
    group, staged through the host (how several ranks share one GPU in the tests; not a production transport)."""

    def __init__(self, handle, keep=None):
        self._L, self._h, self._keep = lib(), handle, keep

    @staticmethod
    def unique_id() -> bytes:
        L = lib()
        buf = (C.c_uint8 * 128)()
        rc = L.cblx_comm_unique_id(buf)
        if rc != OK:
            raise CblxError(rc, L.cblx_last_global_error().decode())
        return bytes(buf)

    @classmethod
    def rccl(cls, uid: bytes, rank: int, world: int, device: int = -1) -> "Comm":
        L = lib()
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        rc = L.cblx_comm_init_rccl(C.byref(h), buf, rank, world, device)
        if rc != OK:
            raise CblxError(rc, L.cblx_last_global_error().decode())
        return cls(h)

    @classmethod
    def over_group(cls, dist, rank: int, world: int, device: int = -1) -> "Comm":
        import numpy as np
        import torch

        dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)

        def view(ptr, n):
            from .sharded import _DeviceArray

            return torch.as_tensor(_DeviceArray(ptr, n, "|u1"), device=dev)

        def all_reduce(_u, vals, n):
            try:
                a = np.ctypeslib.as_array(vals, (n,))
                t = torch.from_numpy(a.astype(np.int64))
                dist.all_reduce(t)
                a[:] = t.numpy().astype(np.uint64)
                return 0
            except Exception:  # noqa: BLE001 - must not unwind into C
                return 1

        def all_to_all(_u, send, recv, per):
            try:
                s = torch.from_numpy(np.ctypeslib.as_array(send, (per * world,)).astype(np.int64))
                r = torch.empty_like(s)
                dist.all_to_all_single(r, s)
                np.ctypeslib.as_array(recv, (per * world,))[:] = r.numpy().astype(np.uint64)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        def exchange(_u, d_src, so, d_dst, ro):
            try:
                so = [int(so[i]) for i in range(world + 1)]
                ro = [int(ro[i]) for i in range(world + 1)]
                src = view(d_src, max(so[world], 1)) if so[world] else None
                dst = view(d_dst, max(ro[world], 1)) if ro[world] else None
                if so[rank + 1] > so[rank]:
                    dst[ro[rank]: ro[rank + 1]].copy_(src[so[rank]: so[rank + 1]])
                works, landing = [], []
                for k in range(1, world):
                    to, frm = (rank + k) % world, (rank - k) % world
                    if so[to + 1] > so[to]:
                        works.append(dist.isend(src[so[to]: so[to + 1]].cpu().contiguous(), to))
                    if ro[frm + 1] > ro[frm]:
                        h = torch.empty(ro[frm + 1] - ro[frm], dtype=torch.uint8)
                        works.append(dist.irecv(h, frm))
                        landing.append((frm, h))
                for w in works:
                    w.wait()
                for frm, h in landing:
                    dst[ro[frm]: ro[frm + 1]].copy_(h)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        cbs = (_ALL_REDUCE_CB(all_reduce), _ALL_TO_ALL_CB(all_to_all), _EXCHANGE_CB(exchange))
        t = Transport(None, *cbs)
        L = lib()
        h = C.c_void_p()
        rc = L.cblx_comm_init_transport(C.byref(h), C.byref(t), rank, world, device)
        if rc != OK:
            raise CblxError(rc, L.cblx_last_global_error().decode())
        return cls(h, keep=cbs)

    @classmethod
    def sim(cls, rank: int, world: int, store_id: int, link_gbps: float = 0.0, device: int = -1) -> "Comm":
        """Rehearsal of one rank of a `world`-GPU job on one GPU (include/cblx.h: cblx_comm_init_sim): ranks 1 .. world-1 record what they
        would send rank 0, rank 0 replays it paced at `link_gbps` GB/s per source rank."""
        L = lib()
        h = C.c_void_p()
        rc = L.cblx_comm_init_sim(C.byref(h), rank, world, device, store_id, link_gbps)
        if rc != OK:
            raise CblxError(rc, L.cblx_last_global_error().decode())
        return cls(h)

    @staticmethod
    def sim_store_free(store_id: int):
        lib().cblx_sim_store_free(store_id)

    PROTOCOLS = {"sorted": 0, "bins": 1, "auto": 2, "replicate": 3}  # CBLX_PROTO_SORTED / _BINS / _AUTO / _REPLICATE (include/cblx.h)

    @staticmethod
    def auto_protocol(world: int) -> str:
        """What "auto" resolves to (the library's rule, restated for callers that shape their slices by it): "replicate" on 2 - 3 ranks
        (one link per pair of GPUs bounds every protocol that ships words: the reads cross instead), "sorted" on 4, "bins" otherwise."""
        return "replicate" if 2 <= world <= 3 else ("sorted" if world == 4 else "bins")

    def protocol_used(self) -> str:
        """What the last sharded insert of this rank ran on ("auto" resolved; "bins" falls back to "sorted" at PREFIX_BITS <= 8)."""
        v = C.c_uint32(0)
        self._L.cblx_comm_protocol_used(self._h, C.byref(v))
        return {0: "sorted", 1: "bins", 3: "replicate"}[v.value]

    def set_protocol(self, name: str):
        """What crosses the links in sharded_insert_seqs_device: "bins" (exchange between the first and the second partition pass),
        "sorted" (full partition on the sender, packed suffixes on the wire) or "auto" (default: sorted on 2 - 4 ranks, bins
        otherwise). The same on every rank."""
        rc = self._L.cblx_comm_set_protocol(self._h, self.PROTOCOLS[name])
        if rc != OK:
            raise CblxError(rc, "cblx_comm_set_protocol")

    def set_recv_groups(self, groups: int):
        """Groups per rank of the "bins" receiver (0 = default: CBLX_RECV_GROUPS or 4; 1 = ungrouped): the data crosses the links
        group-major and the receiver works on group g while g + 1 .. are still on the wire. The same on every rank."""
        rc = self._L.cblx_comm_set_recv_groups(self._h, groups)
        if rc != OK:
            raise CblxError(rc, "cblx_comm_set_recv_groups")

    def groups_used(self) -> int:
        """Groups the last sharded insert of this rank worked through (0: the ungrouped path)."""
        g = C.c_uint32(0)
        self._L.cblx_comm_groups_used(self._h, C.byref(g))
        return g.value

    def groups_fine(self) -> int:
        """... of which sorted 16 prefix bits behind the senders' first pass (FINE bins, PREFIX_BITS > 24: two passes instead of three)."""
        g = C.c_uint32(0)
        self._L.cblx_comm_groups_fine(self._h, C.byref(g))
        return g.value

    def stats(self, reset: bool = False) -> dict:
        st = ExchangeStats()
        self._L.cblx_comm_stats(self._h, C.byref(st), int(reset))
        return {"sent_bytes": st.sent_bytes, "recv_bytes": st.recv_bytes, "messages": st.messages}

    def close(self):
        if getattr(self, "_h", None):
            self._L.cblx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CBL:
    """A set of k-mers backed by an index resident in the HBM of one MI355X.

    `CBL(k, prefix_bits=24)` = `CBL::<K, T, PREFIX_BITS>::new()`; `canonical=True` = `new_canonical()`.
    The reference picks T from K (build.rs:34-41); here the word width follows from k the same way.
    """

    def __init__(self, k: int, prefix_bits: int = 24, canonical: bool = False, device: int = -1, profile: bool = False):
        self._L = lib()
        self.k, self.prefix_bits = k, prefix_bits
        p = Params(k, prefix_bits, int(canonical), device, FLAG_PROFILE if profile else 0, 0)
        h = C.c_void_p()
        rc = self._L.cblx_create(C.byref(p), C.byref(h))
        if rc != OK:
            raise CblxError(rc, self._L.cblx_last_global_error().decode())
        self._h = h

    @classmethod
    def new_canonical(cls, k: int, prefix_bits: int = 24, **kw) -> "CBL":
        return cls(k, prefix_bits, canonical=True, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cblx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != OK:
            raise CblxError(rc, self._L.cblx_last_error(self._h).decode())

    # ---- src/cbl.rs:328-339 ---------------------------------------------------------------------------------
    def insert_seq(self, seq: bytes):
        """Adds all the k-mers of a sequence to the set (enqueued; materialised by the next observer)."""
        self._chk(self._L.cblx_insert_seq(self._h, seq, len(seq)))

    def insert_seqs(self, bases, offsets):
        """The caller loop of examples/cbl.rs:160-163 in one call. numpy uint8 / uint64 (host) arrays."""
        import numpy as np

        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self._L.cblx_insert_seqs(self._h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1))

    def insert_fastx_file(self, path) -> int:
        """`for record in parse_fastx_file(path): insert_seq(record.seq())` (examples/cbl.rs:154-163); returns #records."""
        n = C.c_uint64(0)
        self._chk(self._L.cblx_insert_fastx_file(self._h, os.fsencode(path), C.byref(n)))
        return n.value

    def count_fastx_records(self, path) -> int:
        n = C.c_uint64(0)
        self._chk(self._L.cblx_stage_fastx_blocks(self._h, os.fsencode(path), 0, 0, 1, None, None, None, C.byref(n)))
        return n.value

    def stage_fastx_blocks(self, path, block: int, rank: int, world: int):
        """Parse a FASTA/FASTQ(.gz) file and stage this rank's block-cyclic share in HBM without inserting it. Returns
        (device address of the bases, device address of the n + 1 uint64 offsets, n staged, records in the file); the arrays
        stay valid until stage_release()."""
        pb, po = C.c_void_p(), C.c_void_p()
        n, tot = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.cblx_stage_fastx_blocks(self._h, os.fsencode(path), block, rank, world, C.byref(pb), C.byref(po), C.byref(n), C.byref(tot)))
        return pb.value or 0, po.value or 0, n.value, tot.value

    def stage_fastx_blocks_comm(self, comm, path, block: int = 0, slices: int = 4):
        """The same staging with the parse shared between the ranks of `comm` (every rank reads 1 / world of the file). Returns
        (device address of the bases, of the offsets, n staged, records in the file, block size used)."""
        pb, po = C.c_void_p(), C.c_void_p()
        n, tot, blk = C.c_uint64(0), C.c_uint64(0), C.c_uint64(block)
        self._chk(self._L.cblx_stage_fastx_blocks_comm(self._h, comm._h, os.fsencode(path), C.byref(blk), slices, C.byref(pb), C.byref(po), C.byref(n), C.byref(tot)))
        return pb.value or 0, po.value or 0, n.value, tot.value, blk.value

    def stage_release(self):
        self._chk(self._L.cblx_stage_release(self._h))

    def insert_seqs_device(self, d_bases, d_offsets, n: int):
        """Same with inputs resident in HBM (torch uint8 / int64 CUDA tensors or raw device addresses)."""
        self._chk(self._L.cblx_insert_seqs_device(self._h, _ptr(d_bases), _ptr(d_offsets), n))

    def insert_words_device(self, d_lo, d_hi, n: int):
        """WordSet::insert_batch (src/wordset/mod.rs:187-216) on device-resident words."""
        self._chk(self._L.cblx_insert_words_device(self._h, _ptr(d_lo), _ptr(d_hi), n))

    def seq_words_device(self, d_bases, d_offsets, n: int, d_lo, d_hi, cap: int) -> int:
        """CBL::get_seq_words over every chunk (src/cbl.rs:239-289) into device arrays; returns the word count."""
        nw = C.c_uint64(0)
        self._chk(self._L.cblx_seq_words_device(self._h, _ptr(d_bases), _ptr(d_offsets), n, _ptr(d_lo), _ptr(d_hi), cap, C.byref(nw)))
        return nw.value

    def partition_words_device(self, d_lo, d_hi, n: int, bounds, nd: int, d_out_lo, d_out_hi):
        """Stable partition of device words by destination prefix range; returns the nd run lengths."""
        import numpy as np

        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        counts = np.zeros(nd, dtype=np.uint64)
        self._chk(self._L.cblx_partition_words_device(self._h, _ptr(d_lo), _ptr(d_hi), n, b.ctypes.data if len(b) else None, nd,
                                                      _ptr(d_out_lo), _ptr(d_out_hi), counts.ctypes.data))
        return [int(x) for x in counts]

    def seq_words_partitioned_device(self, d_bases, d_offsets, n: int, bounds, nd: int, d_out_lo, d_out_hi, cap: int):
        """get_seq_words of the sequences, grouped by destination prefix range; returns (n_words, run lengths)."""
        import numpy as np

        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        counts = np.zeros(nd, dtype=np.uint64)
        nw = C.c_uint64(0)
        self._chk(self._L.cblx_seq_words_partitioned_device(self._h, _ptr(d_bases), _ptr(d_offsets), n, b.ctypes.data if len(b) else None, nd,
                                                            _ptr(d_out_lo), _ptr(d_out_hi), cap, counts.ctypes.data, C.byref(nw)))
        return nw.value, [int(x) for x in counts]

    # ---- multi-GPU build, sorted-batch protocol (include/cblx.h) -----------------------------------------------------
    def sorted_batch_begin(self, d_bases, d_offsets, n: int, bounds, nd: int):
        """KRN-1 + full partition; returns (bucket_split, word_split): nd + 1 host entries each."""
        import numpy as np

        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        bs = (C.c_uint64 * (nd + 1))()
        ws = (C.c_uint64 * (nd + 1))()
        self._chk(self._L.cblx_sorted_batch_begin(self._h, _ptr(d_bases), _ptr(d_offsets), n, b.ctypes.data_as(C.POINTER(C.c_uint32)), nd, bs, ws))
        return list(bs), list(ws)

    def sorted_batch_export(self, d_prefix, d_count, d_suffix):
        self._chk(self._L.cblx_sorted_batch_export(self._h, _ptr(d_prefix), _ptr(d_count), _ptr(d_suffix)))

    def insert_sorted_batches_device(self, batches):
        """batches: [(n_buckets, n_words, d_prefix, d_count, d_suffix)] in stream order."""
        arr = (BatchView * max(len(batches), 1))()
        for i, (nb, nw, p, c, s) in enumerate(batches):
            arr[i] = BatchView(nb, nw, _ptr(p) if nb else None, _ptr(c) if nb else None, _ptr(s) if nw else None)
        self._chk(self._L.cblx_insert_sorted_batches_device(self._h, arr, len(batches)))

    # ---- prefix-range sharded indexes (include/cblx.h): one rank's share -------------------------------------------------
    def load_shard_from_file(self, path, rank: int, world: int, bounds=None, sequential: bool = False):
        """This rank's prefix range of an index file. Returns (info dict, bounds as a numpy uint32 array of world-1 values).
        The caller checks info["exact"] and the entry counts over all ranks (cbl_amd.sharded.ShardedIndex does)."""
        import numpy as np

        b = np.ascontiguousarray(bounds, dtype=np.uint32) if bounds is not None else None
        out = np.zeros(max(world - 1, 1), dtype=np.uint32)
        info = ShardInfo()
        self._chk(self._L.cblx_load_shard_from_file(self._h, os.fsencode(path), rank, world, b.ctypes.data if b is not None and len(b) else None,
                                                    int(sequential), out.ctypes.data, C.byref(info)))
        return {n: getattr(info, n) for n, _ in ShardInfo._fields_}, out[: world - 1]

    def resident_split(self, bounds, nd: int):
        """(bucket_split, word_split), nd + 1 entries each: where the bounds cut the resident index."""
        import numpy as np

        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        bs = (C.c_uint64 * (nd + 1))()
        ws = (C.c_uint64 * (nd + 1))()
        self._chk(self._L.cblx_resident_split(self._h, b.ctypes.data if len(b) else None, nd, bs, ws))
        return list(bs), list(ws)

    def resident_export(self, d_prefix, d_count, d_kind, d_suffix):
        self._chk(self._L.cblx_resident_export(self._h, _ptr(d_prefix), _ptr(d_count), _ptr(d_kind), _ptr(d_suffix)))

    def install_buckets_device(self, parts):
        """parts: [(n_buckets, n_words, d_prefix, d_count, d_kind, d_suffix)], ascending prefixes over the concatenation."""
        arr = (BucketView * max(len(parts), 1))()
        for i, (nb, nw, p, c, k, s) in enumerate(parts):
            arr[i] = BucketView(nb, nw, _ptr(p) if nb else None, _ptr(c) if nb else None, _ptr(k) if nb else None, _ptr(s) if nw else None)
        self._chk(self._L.cblx_install_buckets_device(self._h, arr, len(parts)))

    def serialized_body_size(self):
        """(entries, bytes) of the serialized index without its header."""
        ne, nb = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.cblx_serialized_body_size(self._h, C.byref(ne), C.byref(nb)))
        return ne.value, nb.value

    def write_body_at(self, path, file_off: int):
        self._chk(self._L.cblx_write_body_at(self._h, os.fsencode(path), file_off))

    def sharded_insert_seqs_device(self, comm: "Comm", d_bases, d_offsets, n: int, slice_cuts, bounds, bounds_valid: bool):
        """The multi-GPU build step behind the ABI (cblx_sharded_insert_seqs_device): every rank calls it with its shard.
        `bounds`: numpy uint32 array of world - 1 entries, updated in place when bounds_valid is False. Returns True
        (the bounds are valid from now on)."""
        import numpy as np

        cuts = np.ascontiguousarray(slice_cuts, dtype=np.uint64)
        assert bounds.dtype == np.uint32 and bounds.flags["C_CONTIGUOUS"]
        bv = C.c_int(int(bounds_valid))
        self._chk(self._L.cblx_sharded_insert_seqs_device(self._h, comm._h, _ptr(d_bases), _ptr(d_offsets), n, cuts.ctypes.data, len(cuts) - 1,
                                                          bounds.ctypes.data if len(bounds) else None, C.byref(bv)))
        return bool(bv.value)

    def flush(self):
        self._chk(self._L.cblx_flush(self._h))

    # ---- src/cbl.rs:164-177 ---------------------------------------------------------------------------------
    def count(self) -> int:
        v = C.c_uint64(0)
        self._chk(self._L.cblx_count(self._h, C.byref(v)))
        return v.value

    def num_buckets(self) -> int:
        v = C.c_uint64(0)
        self._chk(self._L.cblx_num_buckets(self._h, C.byref(v)))
        return v.value

    def is_empty(self) -> bool:
        v = C.c_int(0)
        self._chk(self._L.cblx_is_empty(self._h, C.byref(v)))
        return bool(v.value)

    def is_canonical(self) -> bool:
        v = C.c_int(0)
        self._chk(self._L.cblx_is_canonical(self._h, C.byref(v)))
        return bool(v.value)

    def contains_seq(self, seq: bytes):
        """For each k-mer of a sequence, True if it is in the set (src/cbl.rs:311-324)."""
        return self.contains_seq_np(seq).tolist()

    def contains_seq_np(self, seq: bytes):
        """contains_seq as a numpy bool array (no per-element Python objects)."""
        import numpy as np

        cap = max(len(seq), 1)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        self._chk(self._L.cblx_contains_seq(self._h, seq, len(seq), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
        return out[: n.value].astype(bool)

    def contains_seqs(self, bases, offsets, flags: bool = True):
        """contains_seq for a batch (bases: uint8 array, offsets: uint64 array of n+1 entries). Returns
        (flags as a numpy bool array or None, k-mers queried, positives)."""
        import numpy as np

        n = len(offsets) - 1
        cap = max(int(offsets[-1] - offsets[0]), 1) if n > 0 else 1
        out = np.empty(cap, dtype=np.uint8) if flags else None
        tot, pos = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.cblx_contains_seqs(self._h, _ptr(bases), _ptr(offsets), max(n, 0), _ptr(out) if flags else None, cap, C.byref(tot), C.byref(pos)))
        return (out[: tot.value].astype(bool) if flags else None), tot.value, pos.value

    def contains_seqs_device(self, d_bases, d_offsets, n: int, d_out=None, cap: int = 0):
        """Device-resident batch; returns (k-mers queried, positives). d_out (uint8 tensor) receives the flags when given."""
        tot, pos = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.cblx_contains_seqs_device(self._h, _ptr(d_bases), _ptr(d_offsets), n, _ptr(d_out), cap, C.byref(tot), C.byref(pos)))
        return tot.value, pos.value

    def query_fastx_file(self, path):
        """`cbl query`: (records, k-mers queried, positives) for a FASTA/FASTQ(.gz) file; the index is not modified."""
        import os

        nrec, tot, pos = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.cblx_query_fastx_file(self._h, os.fsencode(path), C.byref(nrec), C.byref(tot), C.byref(pos)))
        return nrec.value, tot.value, pos.value

    def contains_all(self, seq: bytes) -> bool:
        """True if the set contains all the k-mers of a sequence (src/cbl.rs:293-307)."""
        v = C.c_int(0)
        self._chk(self._L.cblx_contains_all(self._h, seq, len(seq), C.byref(v)))
        return bool(v.value)

    # ---- packed k-mers (IntKmer::to_int(): 2K bits, first base most significant) — src/cbl.rs:219-228,358-361 ------------
    @staticmethod
    def _split_kmers(kmers):
        import numpy as np

        ks = [int(x) for x in kmers]
        lo = np.array([x & 0xFFFFFFFFFFFFFFFF for x in ks], dtype=np.uint64)
        hi = np.array([x >> 64 for x in ks], dtype=np.uint64)
        return lo, hi

    def insert_kmers(self, kmers):
        """n successive `insert` calls; returns their results as a numpy bool array (True = the k-mer was absent)."""
        import numpy as np

        lo, hi = self._split_kmers(kmers)
        out = np.zeros(max(len(lo), 1), dtype=np.uint8)
        self._chk(self._L.cblx_insert_kmers(self._h, _ptr(lo), _ptr(hi), len(lo), _ptr(out)))
        return out[: len(lo)].astype(bool)

    def contains_kmers(self, kmers):
        import numpy as np

        lo, hi = self._split_kmers(kmers)
        out = np.zeros(max(len(lo), 1), dtype=np.uint8)
        self._chk(self._L.cblx_contains_kmers(self._h, _ptr(lo), _ptr(hi), len(lo), _ptr(out)))
        return out[: len(lo)].astype(bool)

    def insert(self, kmer: int) -> bool:
        """Adds a packed k-mer; True if it was absent (src/cbl.rs:226-228)."""
        return bool(self.insert_kmers([kmer])[0])

    def contains(self, kmer: int) -> bool:
        """True if the set contains the packed k-mer (src/cbl.rs:219-221)."""
        return bool(self.contains_kmers([kmer])[0])

    def kmers_np(self):
        """All k-mers of the set in the reference's iteration order as (lo, hi) uint64 arrays (hi is None for K <= 31)."""
        import numpy as np

        n = self.count()
        lo = np.empty(max(n, 1), dtype=np.uint64)
        hi = np.empty(max(n, 1), dtype=np.uint64) if self.k > 31 else None
        got = C.c_uint64(0)
        self._chk(self._L.cblx_export_kmers(self._h, _ptr(lo), _ptr(hi) if hi is not None else None, n, C.byref(got)))
        return lo[: got.value], (hi[: got.value] if hi is not None else None)

    def iter(self):
        """Iterator over the packed k-mers of the set (src/cbl.rs:358-361)."""
        lo, hi = self.kmers_np()
        if hi is None:
            return iter(int(x) for x in lo)
        return iter((int(h) << 64) | int(l) for l, h in zip(lo, hi))

    __iter__ = iter

    # ---- bucket statistics (src/cbl.rs:364-386) ---------------------------------------------------------------------
    def bucket_table_np(self):
        """(prefix, length, kind) arrays of the non-empty buckets, ascending prefixes."""
        import numpy as np

        nb = self.num_buckets()
        prefix = np.empty(max(nb, 1), dtype=np.uint32)
        length = np.empty(max(nb, 1), dtype=np.uint32)
        kind = np.empty(max(nb, 1), dtype=np.uint8)
        got = C.c_uint64(0)
        self._chk(self._L.cblx_bucket_sizes(self._h, _ptr(prefix), _ptr(length), _ptr(kind), nb, C.byref(got)))
        return prefix[: got.value], length[: got.value], kind[: got.value]

    def prefix_load(self) -> float:
        """Proportion of available prefixes used in the set (src/cbl.rs:364-367)."""
        return self.num_buckets() / float(1 << self.prefix_bits)

    def buckets_sizes(self):
        """(prefix, bucket length) pairs, ascending prefixes (src/cbl.rs:370-373)."""
        p, l, _ = self.bucket_table_np()
        return list(zip(p.tolist(), l.tolist()))

    def buckets_size_count(self) -> dict:
        """bucket length -> number of buckets of that length, sorted by length (src/cbl.rs:376-379)."""
        import numpy as np

        _, l, _ = self.bucket_table_np()
        v, c = np.unique(l, return_counts=True)
        return dict(zip(v.tolist(), c.tolist()))

    def buckets_load_repartition(self) -> dict:
        """bucket length -> share of the k-mers held by buckets of that length (src/cbl.rs:382-385)."""
        total = float(max(self.count(), 1))
        return {size: size * n / total for size, n in self.buckets_size_count().items()}

    # ---- src/cbl.rs:127-160 ---------------------------------------------------------------------------------
    def serialize(self) -> bytes:
        """The exact bytes `save_to_file` writes (bincode DefaultOptions + varint)."""
        return self.serialize_np().tobytes()

    def serialized_size(self) -> int:
        n = C.c_uint64(0)
        self._chk(self._L.cblx_serialized_size(self._h, C.byref(n)))
        return n.value

    def serialize_np(self, out=None):
        """serialize() into a numpy uint8 array (`out` if given and large enough); returns the filled view."""
        import numpy as np

        n = self.serialized_size()
        if out is None or out.size < n:
            out = np.empty(max(n, 1), dtype=np.uint8)
        w = C.c_uint64(0)
        self._chk(self._L.cblx_serialize(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), n, C.byref(w)))
        return out[: w.value]

    def save_to_file(self, path):
        self._chk(self._L.cblx_save_to_file(self._h, os.fsencode(path)))

    def load(self, data):
        """Replace the index by a serialized one (bytes, or a numpy uint8 array: no copy is made)."""
        if hasattr(data, "ctypes"):
            self._chk(self._L.cblx_load(self._h, data.ctypes.data, data.size))
        else:
            self._chk(self._L.cblx_load(self._h, data, len(data)))

    @classmethod
    def load_from_file(cls, path, k: int, prefix_bits: int = 24, **kw) -> "CBL":
        """K / PREFIX_BITS are compile-time constants of the reference and not stored in the file."""
        c = cls(k, prefix_bits, **kw)
        c._chk(c._L.cblx_load_from_file(c._h, os.fsencode(path)))
        return c

    # ---- src/cbl.rs:433-449 ---------------------------------------------------------------------------------
    def __ior__(self, other: "CBL") -> "CBL":
        self._chk(self._L.cblx_merge_assign(self._h, other._h))
        return self

    def merge_from(self, a: "CBL", b: "CBL") -> "CBL":
        """self = what `a |= b` would leave in a, with a untouched (b as `|=` leaves it): `self = a.clone(); self |= b` without the copy."""
        self._chk(self._L.cblx_merge_from(self._h, a._h, b._h))
        return self

    # ---- inspection -------------------------------------------------------------------------------------------
    def consts(self) -> dict:
        c = Consts()
        self._chk(self._L.cblx_get_consts(self._h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Consts._fields_}

    def buckets(self):
        """[(prefix, kind, [suffix, ...])] in ascending prefix order; kind 0 = Vec (stored order), 1 = Trie (ascending)."""
        out = []

        def cb(_u, prefix, kind, n, lo, hi):
            if hi:
                out.append((prefix, kind, [lo[i] | (hi[i] << 64) for i in range(n)]))
            else:
                out.append((prefix, kind, [lo[i] for i in range(n)]))
            return 0

        self._chk(self._L.cblx_export_buckets(self._h, BUCKET_CB(cb), None))
        return out

    def checksum(self) -> int:
        """Order-independent 64-bit checksum of the set (sum of a hash of every resident word)."""
        v = C.c_uint64(0)
        self._chk(self._L.cblx_checksum(self._h, C.byref(v)))
        return v.value

    def checksum_words_device(self, d_lo, d_hi, n: int) -> int:
        v = C.c_uint64(0)
        self._chk(self._L.cblx_checksum_words_device(self._h, _ptr(d_lo), _ptr(d_hi), n, C.byref(v)))
        return v.value

    def validate(self, strict: bool = True) -> int:
        """Number of structural violations in the resident buckets (0 = sound)."""
        v = C.c_uint64(0)
        self._chk(self._L.cblx_validate(self._h, int(strict), C.byref(v)))
        return v.value

    def stage_times(self) -> dict:
        """{stage: (ms, launches)} accumulated HIP-event time per pipeline stage (needs profile=True)."""
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        ln = (C.c_uint64 * 16)()
        n = C.c_uint32(0)
        self._chk(self._L.cblx_stage_times(self._h, names, ms, ln, 16, C.byref(n)))
        return {names[i].decode(): (ms[i], ln[i]) for i in range(n.value)}

    def stage_units(self) -> dict:
        """{stage: words} the stage's kernels were given since the last reset, where the pipeline counts them (`|=`); 0 = not counted."""
        names = (C.c_char_p * 16)()
        n = C.c_uint32(0)
        self._chk(self._L.cblx_stage_times(self._h, names, None, None, 16, C.byref(n)))
        un = (C.c_uint64 * 16)()
        self._chk(self._L.cblx_stage_units(self._h, un, 16, C.byref(n)))
        return {names[i].decode(): un[i] for i in range(n.value)}

    def stage_times_reset(self):
        self._chk(self._L.cblx_stage_times_reset(self._h))

    def kmers_inserted(self) -> int:
        v = C.c_uint64(0)
        self._chk(self._L.cblx_kmers_inserted(self._h, C.byref(v)))
        return v.value

    def fine_builds(self) -> int:
        """Batches built through the FINE-bins route (PREFIX_BITS > 24, empty index, CBLX_FINE_MIN k-mers or more)."""
        v = C.c_uint64(0)
        self._chk(self._L.cblx_fine_builds(self._h, C.byref(v)))
        return v.value

    def trim(self):
        self._chk(self._L.cblx_trim(self._h))

    def clear(self):
        """Back to `CBL::new()`: empty set, cached device workspace kept."""
        self._chk(self._L.cblx_clear(self._h))
