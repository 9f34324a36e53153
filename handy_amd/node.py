"""Node-level data parallelism by utterance: one process (one independent engine) per GPU, no tensor
parallelism; the only collective is the gather of per-utterance token records at the end of a pass.
Handy has no multi-GPU path (one engine behind one mutex, reference
src-tauri/src/managers/transcription.rs:39, :383); this module is what a bulk driver runs on top of the
batch entry point."""
from __future__ import annotations

import base64
import glob
import json
import os
import zlib
from typing import Dict, List, Optional, Sequence, Set

import numpy as np


def shard_indices(n_total: int, rank: int, world: int) -> np.ndarray:
    """Contiguous, balanced shard of utterance indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return np.arange(start, start + base + (1 if rank < rem else 0), dtype=np.int64)


def pack_token_records(index: Sequence[int], tokens: Sequence[Sequence[int]], width: int) -> np.ndarray:
    """Fixed-size records [utterance index, n_tokens, tokens..., -1 padding] so the gather is one dense tensor."""
    rec = np.full((len(index), width + 2), -1, dtype=np.int32)
    for i, (u, t) in enumerate(zip(index, tokens)):
        t = list(t)[:width]
        rec[i, 0], rec[i, 1] = u, len(t)
        rec[i, 2:2 + len(t)] = t
    return rec


def gather_token_records(rec: np.ndarray, rows_per_rank: int, device=None) -> Optional[np.ndarray]:
    """Gathers every rank's records on rank 0 (RCCL when the default group is 'nccl', gloo on CPU).
    Ranks pad to `rows_per_rank` rows; rank 0 returns the records sorted by utterance index, others None."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    pad = np.full((rows_per_rank, rec.shape[1]), -1, dtype=np.int32)
    pad[:len(rec)] = rec
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return None
    allr = torch.cat(out).cpu().numpy()
    allr = allr[allr[:, 0] >= 0]
    return allr[np.argsort(allr[:, 0], kind="stable")]


class BatchQueue:
    """A shared queue of batch numbers 0 .. n_batches-1 for the ranks of one job: every rank pulls the next batch when it is
    free, so ranks whose utterances decode for longer simply pull fewer (dynamic balance by utterance batch; no data moves
    between GPUs). The counter is one atomic fetch-add on the job's rendezvous store (torch.distributed's TCP store), not a
    collective; without a process group it is a local counter."""

    def __init__(self, n_batches: int, name: str = "wmi_batch_queue"):
        import torch.distributed as dist
        self.n_batches, self.key, self._local = int(n_batches), name, 0
        self.store = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = _get_default_store()

    def pull(self) -> Optional[int]:
        if self.store is not None:
            i = int(self.store.add(self.key, 1)) - 1
        else:
            i, self._local = self._local, self._local + 1
        return i if i < self.n_batches else None

    def __iter__(self):
        while True:
            i = self.pull()
            if i is None:
                return
            yield i


def gather_ragged_records(rec: np.ndarray, device=None) -> Optional[np.ndarray]:
    """gather_token_records for ranks that hold different numbers of records (what a pulled queue leaves): the counts are
    exchanged first, every rank pads to the largest. Rank 0 gets all records sorted by utterance index."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec[np.argsort(rec[:, 0], kind="stable")]
    n = torch.tensor([len(rec)], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    counts = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, n)
    return gather_token_records(rec, rows_per_rank=max(1, max(int(c.item()) for c in counts)), device=device)


class ResultLog:
    """Append-only log of the finished batches of one bulk job, one file per rank, so that a job of thousands of utterances
    that dies at batch 39 of 40 resumes at batch 39 instead of losing the run (Handy itself only resumes model downloads:
    reference src-tauri/src/managers/model.rs:361-405; it has no bulk job to resume).

    One line per finished batch: {"job", "batch", "shape", "rec" (base64 of the int32 records), "crc"}, written with one
    write call and fsync-ed before the batch counts as done. A torn or corrupt last line -- the crash happened inside the
    write -- fails its JSON parse or its checksum and is ignored: that batch is simply run again. Lines of another job (a
    different `job` key) are ignored as well. A rank owns its own file; files of ranks beyond the current world size (a resume
    with fewer GPUs) are owned by rank 0, so every logged record is handed to the final gather exactly once."""

    def __init__(self, directory: str, job: str, rank: int = 0, world: int = 1, resume: bool = True):
        self.dir, self.job, self.rank, self.world = directory, job, int(rank), int(world)
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, f"rank{self.rank:04d}.jsonl")
        if not resume and os.path.exists(self.path):
            os.remove(self.path)
        self._batches: Dict[int, np.ndarray] = {}     # finished batches whose records this rank owns
        self._done: Set[int] = set()                  # finished batches of every rank
        for f in sorted(glob.glob(os.path.join(directory, "rank*.jsonl"))) if resume else []:      # a fresh run reads nobody's old lines
            try:
                owner = int(os.path.basename(f)[4:8])
            except ValueError:
                continue
            mine = owner == self.rank or (self.rank == 0 and owner >= self.world)
            for b, rec in self._read(f):
                if b in self._done and not (mine and b not in self._batches):
                    continue
                self._done.add(b)
                if mine:
                    self._batches[b] = rec
        torn = False
        if os.path.exists(self.path) and os.path.getsize(self.path) > 0:
            with open(self.path, "rb") as fh:
                fh.seek(-1, os.SEEK_END)
                torn = fh.read(1) != b"\n"
        self._fh = open(self.path, "ab")
        if torn:                                       # the crash left half a line: the next record starts on a line of its own
            self._fh.write(b"\n")

    def _read(self, path: str):
        try:
            with open(path, "rb") as fh:
                lines = fh.read().split(b"\n")
        except OSError:
            return
        for line in lines:
            if not line.strip():
                continue
            try:
                d = json.loads(line)
                raw = base64.b64decode(d["rec"])
                if d["job"] != self.job or zlib.crc32(raw) != d["crc"]:
                    continue
                yield int(d["batch"]), np.frombuffer(raw, dtype=np.int32).reshape(d["shape"]).copy()
            except (ValueError, KeyError, TypeError):
                continue                              # a torn line: the batch it was about runs again

    def done(self) -> Set[int]:
        return set(self._done)

    def records(self) -> List[np.ndarray]:
        """Records of the logged batches this rank owns, in batch order."""
        return [self._batches[b] for b in sorted(self._batches)]

    def append(self, batch: int, rec: np.ndarray) -> None:
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        raw = rec.tobytes()
        line = json.dumps({"job": self.job, "batch": int(batch), "shape": list(rec.shape), "rec": base64.b64encode(raw).decode("ascii"),
                           "crc": zlib.crc32(raw)}).encode("ascii") + b"\n"
        self._fh.write(line)
        self._fh.flush()
        os.fsync(self._fh.fileno())
        self._done.add(int(batch))
        self._batches[int(batch)] = rec

    def close(self) -> None:
        self._fh.close()


class NativeNode:
    """The C-level node driver (include/wmi.h, wmi_node_*): one process, one engine and one worker thread per GPU pulling
    batches from an atomic counter, result records gathered over RCCL. `devices=None`: every visible GPU."""

    def __init__(self, path: str, devices: Optional[Sequence[int]] = None, batch: int = 256, max_decoders: int = 1, enc_chunk: int = 0):
        import ctypes as C
        from . import whisper_rs as wr
        self._C, self._wr = C, wr
        self.L = wr.load_library()
        L = self.L
        L.wmi_node_init.restype = C.c_void_p
        L.wmi_node_init.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
        L.wmi_node_free.restype = None; L.wmi_node_free.argtypes = [C.c_void_p]
        L.wmi_node_n_devices.restype = C.c_int; L.wmi_node_n_devices.argtypes = [C.c_void_p]
        L.wmi_node_state.restype = C.c_void_p; L.wmi_node_state.argtypes = [C.c_void_p, C.c_int]
        L.wmi_node_run.restype = C.c_int
        L.wmi_node_run.argtypes = [C.c_void_p, wr.whisper_full_params, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(wr.wmi_node_record)]
        L.wmi_node_last_gather.restype = C.c_int; L.wmi_node_last_gather.argtypes = [C.c_void_p]
        L.wmi_node_last_batches.restype = C.c_int; L.wmi_node_last_batches.argtypes = [C.c_void_p, C.c_int]
        devs = (C.c_int * len(devices))(*devices) if devices else None
        self.h = L.wmi_node_init(path.encode(), devs, len(devices) if devices else 0, batch, max_decoders, enc_chunk)
        if not self.h:
            raise wr.WhisperError("InitError: wmi_node_init failed")
        self.n_devices = L.wmi_node_n_devices(self.h)

    def set_force_tokens(self, n: int):
        for i in range(self.n_devices):
            self.L.wmi_set_force_tokens(self.L.wmi_node_state(self.h, i), int(n))

    def run(self, params, pcm_list: Sequence[np.ndarray]):
        """whisper_full over the clips, spread over the node's engines. Returns the list of wmi_node_record (utterance order)."""
        C = self._C
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in pcm_list]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ns = (C.c_int * len(arrs))(*[a.size for a in arrs])
        recs = (self._wr.wmi_node_record * max(len(arrs), 1))()
        rc = self.L.wmi_node_run(self.h, params.c, ptrs, ns, len(arrs), recs)
        if rc != 0:
            raise self._wr.WhisperError("GenericError: wmi_node_run", rc)
        return [recs[i] for i in range(len(arrs))]

    def last_gather(self) -> str:
        return "rccl" if self.L.wmi_node_last_gather(self.h) == 1 else "host"

    def batches_per_engine(self) -> List[int]:
        return [self.L.wmi_node_last_batches(self.h, i) for i in range(self.n_devices)]

    def free(self):
        if self.h:
            self.L.wmi_node_free(self.h)
            self.h = None
