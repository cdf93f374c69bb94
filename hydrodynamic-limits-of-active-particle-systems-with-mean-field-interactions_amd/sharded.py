"""Particle-index sharding across GPUs: one process per GPU, ONE all-gather of proposal bytes per step.

Protocol of one step (identical on every rank):
    engine.propose()          rank r evaluates the all-pairs sums + draws proposals for ITS shard only
                              (N^2 / P pairs) and writes them into block r of the proposal buffer
    all_gather(proposals)     1 byte per particle; in place
    engine.commit()           every rank applies ALL proposals redundantly (O(N), deterministic) and so
                              holds the identical full state again
There is no second round for the exclusion conflicts: the commit rule is a pure function of
(state, proposals).

`engine` is anything with propose() / commit() / exchange tensor; on the GPU it is `HipEngine` (kernels
behind the C ABI, torch.distributed backend "nccl" = RCCL over xGMI).  The CPU tests drive the same
class with the oracle as engine over "gloo".
"""
from __future__ import annotations


def shard_length(n_particles: int, world: int) -> int:
    """Slots per rank: whole 256-slot groups (must match aps_create in csrc/aps_hip.hip)."""
    per_rank = (n_particles + world - 1) // world
    return max(256, (per_rank + 255) // 256 * 256)


class ShardedStepper:
    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        buf, off, mine = engine.exchange()
        assert buf.numel() == mine * self.world and off == mine * self.rank
        self.buf, self.mine = buf, buf[off:off + mine]

    def step(self, nsteps=1):
        for _ in range(int(nsteps)):
            self.engine.propose()
            if self.world > 1:
                self.dist.all_gather_into_tensor(self.buf, self.mine, group=self.group)
            self.engine.commit()


class HipEngine:
    """The HIP stepper as a sharding engine.  The proposal buffer is a torch CUDA tensor handed to the
    library by raw pointer; kernels run on torch's current stream, so the collective is ordered with
    them without extra synchronisation."""

    def __init__(self, handle, device):
        import torch
        self.torch = torch
        self.h = handle
        _, total, off, mine = handle.exchange_buffer()
        self.buf = torch.zeros(total, dtype=torch.uint8, device=device)
        self.off, self.mine_bytes = off, mine
        handle.set_stream(torch.cuda.current_stream(device).cuda_stream)
        handle.bind_exchange_buffer(self.buf.data_ptr(), total)

    def exchange(self):
        return self.buf, self.off, self.mine_bytes

    def propose(self):
        self.h.propose()

    def commit(self):
        self.h.commit()
