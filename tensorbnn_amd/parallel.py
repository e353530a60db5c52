"""Independent chains, one per GPU, one process per GPU (torch.distributed; the
"nccl" backend is RCCL on ROCm).  HMC chains never exchange anything on the
data path; the only collective is the all-gather of the sampled state
(theta, eta) at checkpoint time.  The reference has no
distributed code at all -- this module is new.

Works with any torch.distributed backend: "nccl" on the GPU box, "gloo" in the
CPU tests (world_size 2).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def gather_samples(sample: torch.Tensor) -> torch.Tensor:
    """all-gather one sampled state per rank -> [world, P+H] (chain-major)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sample.reshape(1, -1).clone()
    world = dist.get_world_size()
    out = torch.empty(world * sample.numel(), dtype=sample.dtype, device=sample.device)
    dist.all_gather_into_tensor(out, sample.contiguous().reshape(-1))
    return out.reshape(world, -1)


class SampleGatherer:
    """Collects the chain-major gathered samples on every rank and lets rank 0
    write one reference-format folder per chain (chain<c>/).  With a native communicator (`comm` = make_comm(chain)) the
    gather is the C ABI's own RCCL all-gather on the chain's stream (tbnn_gather_samples) -- the route bench.py times;
    without one it goes through torch.distributed (any backend: the gloo CPU tests drive it with fake chains)."""

    def __init__(self, P, H, device=None, comm=None):
        self.P, self.H = P, H
        self.comm = comm
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.buf = torch.empty(P + H, dtype=torch.float32, device=self.device)
        self.samples = []            # list of [world, P+H] host arrays

    def __call__(self, chain, iter_):
        """train(gather=...) hook: export the chain's (theta, eta) on the device and all-gather it."""
        if self.comm is not None:
            self.samples.append(chain.gather_samples(self.comm))
            return
        if self.buf.is_cuda:
            chain.export_sample_device(self.buf.data_ptr())
        else:   # CPU tests drive this path with fake chains
            self.buf.copy_(torch.from_numpy(np.concatenate([chain.get_state(), chain.get_hypers()])))
        self.samples.append(gather_samples(self.buf).cpu().numpy())

    def stacked(self):
        """[n_samples, world, P+H]"""
        return np.stack(self.samples) if self.samples else np.zeros((0, 1, self.P + self.H), np.float32)


def write_chain_folders(root, stacked, shapes, layer_names, n_hyper):
    """rank 0: one folder per chain in the reference's sample format
    (writer network.py:545-663; reader predictor.py:43-113), one file set."""
    n_samp, world, _ = stacked.shape
    for c in range(world):
        d = os.path.join(root, f"chain{c}")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "architecture.txt"), "wb") as f:
            for name in layer_names:
                f.write((name + "\n").encode("utf-8"))
        o = 0
        for n, shp in enumerate(shapes):
            size = int(np.prod(shp))
            with open(os.path.join(d, f"{n}.0.txt"), "wb") as f:
                for s in range(n_samp):
                    np.savetxt(f, stacked[s, c, o:o + size].reshape(shp))
            o += size
        with open(os.path.join(d, "hypers0.txt"), "wb") as f:
            for s in range(n_samp):
                np.savetxt(f, stacked[s, c, o:o + n_hyper].reshape(-1, 1))
        with open(os.path.join(d, "summary.txt"), "wb") as f:
            for shp in shapes:
                f.write((" ".join(str(x) for x in shp) + "\n").encode("utf-8"))
            f.write(f"{n_samp} 1 {len(shapes)}\n".encode("utf-8"))
            f.write(str(n_hyper).encode("utf-8"))


# ---------------------------------------------------------------------------------------------
# native RCCL communicator (include/tbnn.h, tbnn_comm_*) and the row-sharded single chain
# ---------------------------------------------------------------------------------------------
def row_block(n: int, rank: int, world: int):
    """rows [lo, hi) of rank `rank`: contiguous blocks, multiples of 16 rows (one MFMA row tile) except the last"""
    tiles = (n + 15) // 16
    per = (tiles + world - 1) // world
    lo = min(n, rank * per * 16)
    hi = min(n, (rank + 1) * per * 16)
    return lo, hi


def make_comm(chain):
    """One native communicator per chain: rank 0 draws the RCCL unique id, torch.distributed (any backend)
    carries it to the other ranks, every rank joins on its chain's device."""
    from . import _native as nat
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    box = [nat.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return nat.Comm(chain, world, rank, box[0])


def shard_rows(chain, X, Y, comm):
    """Row-sharded single chain: every rank keeps rows row_block(n, rank, world) of (X, Y) and
    the same (theta, eta, seed, chain_id); the native library all-reduces the data-term gradient (P floats) and
    the likelihood statistic over RCCL/xGMI after every fused pass."""
    n = X.shape[0]
    lo, hi = row_block(n, comm.rank, comm.world)
    if hi <= lo:
        raise ValueError(f"rank {comm.rank} of {comm.world} would hold no rows (n={n})")
    chain.set_data(np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(Y[lo:hi]))
    chain.set_row_shard(comm, n)
    return lo, hi
