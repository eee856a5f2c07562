"""Tensor-parallel plumbing for the latency-oriented multi-GPU mode (second form).

Throughput mode needs none of this: generations are independent, so `bench.py` runs one replica per GPU with no
data-path collective. The classes here serve the OTHER sharding the north-star names — one generation
spread over several GPUs of a node:

  * Llama decoder, Megatron layout: q/k/v/gate/up column-parallel (heads / FFN rows split), o/down row-parallel, ONE
    all-reduce of the fp32 residual stream after each row-parallel GEMM (2 per layer), lm_head vocab-parallel with an
    all-gather of the [G, V/tp] logits in front of the (replicated, deterministic) greedy rule.
  * SDXL UNet, CFG-parallel: the nb guidance branches of one denoise step run on nb ranks, one all-gather of the
    [G, H·W, 4] eps per step (256 KB at 1024 px) in front of the replicated CFG + Euler update.

`Comm` is the only thing the model code sees. `TorchDistComm` maps it to torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests); `IpcComm` runs the small decode-step collectives as one graph-capturable
kernel launch over hipIpc-mapped peer buffers (csrc/comm.hip). `ThreadComm` runs `world` virtual ranks as threads of ONE process on ONE
GPU — the single-GPU box the kernels are validated on cannot host a real multi-rank RCCL group, and the sharding
arithmetic (which rows / columns / heads each rank owns, where the reductions sit) is what has to be proven.
"""
import threading

import torch


class Comm:
    """Single-rank communicator: every collective is the identity."""
    rank, world = 0, 1
    graph_safe = True     # collectives may be captured into a HIP graph

    def all_reduce(self, t):
        return t

    def all_gather(self, t):
        """t: [...] → [world, ...] (rank-major)."""
        return t.unsqueeze(0)

    def all_gather_async(self, t):
        """Starts the all-gather and returns a handle whose ``wait()`` yields the gathered tensor: the collective runs on
        the communicator's own (side) stream and is fenced against the caller's stream by events, so kernels enqueued
        between the call and ``wait()`` overlap with the exchange."""
        return _Done(self.all_gather(t))

    def barrier(self):
        pass

    def halo_exchange(self, edges):
        """edges: [2, ...] = (my FIRST row, my LAST row) of a row-sharded slab → [2, ...] = (LAST row of rank - 1, FIRST row of
        rank + 1); zeros where there is no neighbour. Default: an all-gather of everybody's edge rows (any backend)."""
        allr = self.all_gather(edges.contiguous())                               # [world, 2, ...]
        out = torch.zeros_like(edges)
        if self.rank > 0:
            out[0] = allr[self.rank - 1, 1]
        if self.rank < self.world - 1:
            out[1] = allr[self.rank + 1, 0]
        return out

    def require_capacity(self, n_floats):
        """Called by a model that will CAPTURE collectives of up to n_floats fp32 elements into a HIP graph: a communicator
        that could not run such a payload inside a capture raises here, at load time, instead of failing mid-capture."""

    def check(self):
        """Raises if an earlier collective failed asynchronously (bounded poll expired). Host sync; no-op by default."""


class _Done:
    def __init__(self, value):
        self._v = value

    def wait(self):
        return self._v


class TorchDistComm(Comm):
    """torch.distributed process group (RCCL on GPUs, gloo on CPU). Collectives are enqueued by c10d on its own stream
    and ordered against the caller's current stream by events; the model code needs no explicit synchronisation."""
    graph_safe = False    # not validated under HIP-graph capture on this pool → TP decode runs eager launches

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t):
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            self._dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)   # flat: any backend
        else:                                          # gloo: list form
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self._dist.all_gather(parts, t.contiguous(), group=self.group)
            out = torch.stack(parts, dim=0)
        return out

    def all_gather_async(self, t):
        """c10d enqueues the RCCL all-gather on its internal stream after an event on the caller's stream; Work.wait()
        makes the caller's stream wait for it — the side-stream + event-fence overlap of SURVEY.md §8b."""
        if not t.is_cuda:
            return _Done(self.all_gather(t))
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        work = self._dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group, async_op=True)

        class _H:
            def wait(_s):
                work.wait()
                return out
        return _H()

    def barrier(self):
        self._dist.barrier(group=self.group)


class IpcComm(Comm):
    """One process per GPU; all-reduce / all-gather of small fp32 payloads as ONE kernel launch per rank over peer-mapped
    staging buffers (csrc/comm.hip: hipIpcMemHandle, epoch flags, rank-ordered reduction → the same bits on every rank, and
    the same bits as ThreadComm's `parts[0] + parts[1] + …`). Graph-capturable: the kernel carries its epoch in device memory,
    so the tensor-parallel decode step can be captured and replayed with its 80 all-reduces inside.

    `bootstrap` is any torch.distributed group (gloo is enough): it only carries the 64-byte IPC handles at construction
    and serves payloads above `cap_floats` (RCCL's ring is the right tool there). Peers may be other GPUs of the node
    (xGMI peer access) or — how the 1-GPU test pool exercises the protocol — other processes on the SAME GPU."""
    graph_safe = True

    CHUNK = 4096                               # floats per workgroup (SX_ONESHOT_CHUNK in include/seedx_hip.h)

    def __init__(self, bootstrap=None, cap_floats=131072, device=None, max_spin=0, graph_safe=True, neighbor_only=False,
                 self_test=True):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._dist, self.group = dist, bootstrap
        # graph_safe=False: for callers whose payloads also take the bootstrap-group fallback (the row-sharded UNet: halo rows
        # fit the one-shot kernel, the K|V gathers and fp64 GroupNorm sums go to RCCL) — an RCCL call cannot be captured
        self.graph_safe = bool(graph_safe)
        # neighbor_only: this communicator ONLY runs halo exchanges (sx_oneshot_args.mode 1: a rank signals / waits for rank ± 1);
        # its epochs must not interleave with all-rank collectives, so halo_exchange() of an ordinary IpcComm builds one lazily
        self.neighbor_only, self._halo = bool(neighbor_only), None
        self.rank, self.world = dist.get_rank(bootstrap), dist.get_world_size(bootstrap)
        self.cap, self.max_spin = -(-int(cap_floats) // self.CHUNK) * self.CHUNK, int(max_spin)
        nchunk = self.cap // self.CHUNK
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._lib = lib = _lib.load()
        with torch.cuda.device(self.device):
            own = []
            for nbytes in (2 * self.cap * 4, nchunk * self.world * 4):                # staging (2 slots), flags [chunk][world]
                p = C.c_void_p()
                _lib.check(lib.sx_comm_alloc(C.byref(p), nbytes), "sx_comm_alloc")
                own.append(p.value)
            self._own = own
            handles = []
            for p in own:
                h = C.create_string_buffer(64)
                _lib.check(lib.sx_ipc_export(p, h), "sx_ipc_export")
                handles.append(h.raw)
            gathered = [None] * self.world
            dist.all_gather_object(gathered, handles, group=bootstrap)
            self._opened = []
            stage_ptrs, flag_ptrs = [], []
            for r in range(self.world):
                if r == self.rank:
                    stage_ptrs.append(own[0]); flag_ptrs.append(own[1])
                    continue
                ptrs = []
                for h in gathered[r]:
                    q = C.c_void_p()
                    _lib.check(lib.sx_ipc_open(h, C.byref(q)), "sx_ipc_open")
                    ptrs.append(q.value)
                    self._opened.append(q.value)
                stage_ptrs.append(ptrs[0]); flag_ptrs.append(ptrs[1])
            self._stage = torch.tensor(stage_ptrs, dtype=torch.int64, device=self.device)
            self._flags = torch.tensor(flag_ptrs, dtype=torch.int64, device=self.device)
            self._epoch = torch.zeros(nchunk, dtype=torch.int32, device=self.device)
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
            torch.cuda.synchronize()
        dist.barrier(group=bootstrap)          # every rank has opened every handle before the first collective
        # First contact with the peers' memory (other GPUs over xGMI, or other processes on this GPU): prove the protocol before any
        # model uses it. A failure here (peer stores not visible, a flag that never arrives, wrong sums) must not become wrong tokens or
        # a 16-s spin inside generate(): the communicator then routes EVERY collective to the bootstrap group (RCCL / gloo) and says so.
        self.fallback_reason = None
        if self_test:
            self._self_test()

    SELF_TEST_SPIN = 1 << 22      # bounded poll of the self-test's launches (~ a few seconds at worst, then a status word, not a hang)

    def _self_test(self):
        import warnings
        ok, why = True, ""
        spin, self.max_spin = self.max_spin, self.SELF_TEST_SPIN
        try:
            with torch.cuda.device(self.device):
                n = min(self.cap, 3 * self.CHUNK + 17 * 4)               # several chunks + a ragged tail
                idx = torch.arange(n, device=self.device, dtype=torch.float32)
                if self.neighbor_only:
                    for ep in range(4):                                   # both staging slots, twice
                        send = torch.stack([idx[:2048] + 1000.0 * self.rank + ep, -(idx[:2048] + 1000.0 * self.rank + ep)])   # [first | last]
                        got = self.halo_exchange(send)
                        exp0 = -(idx[:2048] + 1000.0 * (self.rank - 1) + ep) if self.rank > 0 else None
                        exp1 = idx[:2048] + 1000.0 * (self.rank + 1) + ep if self.rank < self.world - 1 else None
                        torch.cuda.synchronize()
                        if int(self._status.item()):
                            ok, why = False, f"halo epoch {ep}: a neighbour's flag never arrived"
                            break
                        if (exp0 is not None and not torch.equal(got[0], exp0)) or (exp1 is not None and not torch.equal(got[1], exp1)):
                            ok, why = False, f"halo epoch {ep}: neighbour rows arrived with wrong contents"
                            break
                else:
                    for ep in range(4):
                        t = (idx % 7 + 1.0) * (self.rank + 1) + ep       # small integers: the rank-ordered fp32 sum is exact
                        self._launch(t)
                        exp = (idx % 7 + 1.0) * (self.world * (self.world + 1) // 2) + ep * self.world
                        torch.cuda.synchronize()
                        if int(self._status.item()):
                            ok, why = False, f"all-reduce epoch {ep}: a peer's flag never arrived within {self.SELF_TEST_SPIN} polls"
                            break
                        if not torch.equal(t, exp):
                            ok, why = False, f"all-reduce epoch {ep}: wrong sum (max abs deviation {float((t - exp).abs().max()):.3g})"
                            break
                    if ok:
                        mine = idx[:1024] + 4096.0 * self.rank
                        out = torch.empty((self.world, 1024), dtype=torch.float32, device=self.device)
                        self._launch(mine.contiguous(), gather_out=out)
                        torch.cuda.synchronize()
                        exp = idx[:1024][None, :] + 4096.0 * torch.arange(self.world, device=self.device, dtype=torch.float32)[:, None]
                        if int(self._status.item()) or not torch.equal(out, exp):
                            ok, why = False, "all-gather: wrong contents or a missing peer"
        except RuntimeError as e:            # a launch error is a failed self-test too
            ok, why = False, f"launch failed: {e}"
        self.max_spin = spin
        verdicts = [None] * self.world
        self._dist.all_gather_object(verdicts, (bool(ok), why), group=self.group)
        bad = [(r, w) for r, (o, w) in enumerate(verdicts) if not o]
        if bad:
            self.fallback_reason = "; ".join(f"rank {r}: {w}" for r, w in bad)
            self.graph_safe = False
            warnings.warn(f"IpcComm rank {self.rank}: the one-shot IPC collectives failed their start-up self-test ({self.fallback_reason}); "
                          f"every collective of this communicator now runs on the bootstrap process group instead (not graph-capturable)")

    def halo_exchange(self, edges):
        """Neighbour-only exchange over the peer-mapped staging buffers: ONE launch, 2 x row bytes in and out, no all-gather of
        every rank's edge rows (sx_oneshot_args.mode 1). 16-bit rows travel as 32-bit words."""
        if not self.neighbor_only:
            if self._halo is None:                       # collective: every rank reaches its first halo exchange together
                self._halo = IpcComm(self.group, cap_floats=self.cap, device=self.device, max_spin=self.max_spin,
                                     graph_safe=self.graph_safe, neighbor_only=True, self_test=self.fallback_reason is None)
                if self.fallback_reason is not None:
                    self._halo.fallback_reason = self.fallback_reason
            return self._halo.halo_exchange(edges)
        e = edges.contiguous()
        assert e.shape[0] == 2 and e.is_cuda
        send = torch.stack([e[1], e[0]], dim=0).contiguous()                    # [my last row | my first row]
        nbytes = send.numel() * send.element_size()
        if self.fallback_reason is not None or nbytes % 16 or nbytes // 4 > self.cap or (send.storage_offset() * send.element_size()) % 8:
            return self._halo_fallback(edges)
        # zeros, not empty: a timed-out exchange returns before writing the neighbour rows (sticky status, check()); the row-sharded
        # UNet must then see zero padding rows, not uninitialised memory, until its per-forward check() raises
        out = torch.zeros_like(send)
        self._launch(send.view(-1).view(torch.float32), gather_out=out.view(-1).view(torch.float32), mode=1)
        return out

    def _halo_fallback(self, edges):
        self._fallback_guard(edges, "halo_exchange")
        parts = [torch.empty_like(edges) for _ in range(self.world)]
        self._dist.all_gather(parts, edges.contiguous(), group=self.group)
        out = torch.zeros_like(edges)
        if self.rank > 0:
            out[0] = parts[self.rank - 1][1]
        if self.rank < self.world - 1:
            out[1] = parts[self.rank + 1][0]
        return out

    def _launch(self, t, gather_out=None, mode=0):
        import ctypes as C
        from . import _lib
        assert mode == 1 or not self.neighbor_only, "a neighbor_only IpcComm runs halo exchanges only"
        a = _lib.OneshotArgs()
        a.mode = mode
        a.data, a.gather_out = t.data_ptr(), (gather_out.data_ptr() if gather_out is not None else None)
        a.stage, a.flags = self._stage.data_ptr(), self._flags.data_ptr()
        a.epoch, a.status = self._epoch.data_ptr(), self._status.data_ptr()
        a.n, a.cap, a.rank, a.world, a.max_spin, a.chunk = t.numel(), self.cap, self.rank, self.world, self.max_spin, self.CHUNK
        _lib.check(self._lib.sx_allreduce_oneshot(C.byref(a), torch.cuda.current_stream().cuda_stream), "sx_allreduce_oneshot")

    def _fits(self, t):
        return self.fallback_reason is None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and 0 < t.numel() <= self.cap

    def require_capacity(self, n_floats):
        if self.fallback_reason is not None:
            return                              # nothing is captured any more: graph_safe is False
        if self.graph_safe and n_floats > self.cap:
            raise RuntimeError(f"IpcComm: a captured collective of {n_floats} floats does not fit the staging capacity {self.cap}; "
                               f"construct it with cap_floats >= {n_floats}")

    def _fallback_guard(self, t, what):
        # the bootstrap-group path is a host-driven RCCL / gloo call: inside a HIP-graph capture it is either a capture error
        # or work that silently is not part of the replayed graph
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"IpcComm.{what}: payload {tuple(t.shape)} {t.dtype} does not fit the one-shot kernel "
                               f"(fp32, contiguous, <= {self.cap} elements) and the fallback cannot run under graph capture")

    def all_reduce(self, t):
        if self._fits(t):
            self._launch(t)
        else:                                  # large / non-fp32 payloads: the bootstrap group's ring
            self._fallback_guard(t, "all_reduce")
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t):
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        nbytes = t.numel() * t.element_size()
        if self._fits(t):
            self._launch(t, gather_out=out)
        elif self.fallback_reason is None and t.is_cuda and t.dtype != torch.float32 and nbytes % 4 == 0 and 0 < nbytes // 4 <= self.cap:
            # a gather moves bits, it does no arithmetic: any dtype travels as 32-bit words (the row-sharded UNet's 16-bit conv
            # halo rows, int32 ids, ...) through the same one-shot kernel
            if (t.storage_offset() * t.element_size()) % 4:
                t = t.clone()                    # a 16-bit view starting on an odd element: re-base it on a 4-byte boundary
            self._launch(t.view(-1).view(torch.float32), gather_out=out.view(-1).view(torch.float32))
        else:
            self._fallback_guard(t, "all_gather")
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self._dist.all_gather(parts, t, group=self.group)
            out = torch.stack(parts, dim=0)
        return out

    def check(self):
        """Raises if a collective gave up waiting for a peer: the kernel then returned WITHOUT reducing, so everything
        computed since is wrong. The status word is sticky (first failed epoch) and the communicator stays failed — the
        slot-reuse argument of csrc/comm.hip no longer holds after a missed epoch. One 4-byte read-back; the generate loop
        calls it next to its per-token read-back."""
        st = int(self._status.item()) if self.fallback_reason is None else 0     # (after a failed self-test the kernel is never used again)
        if st:
            raise RuntimeError(f"IpcComm rank {self.rank}: a peer did not arrive in epoch {st} (bounded poll expired); "
                               f"results since then are invalid and this communicator must be rebuilt")
        if self._halo is not None:            # the neighbour-only child communicator of halo_exchange() has a status word of its own
            self._halo.check()

    def barrier(self):
        self._dist.barrier(group=self.group)

    def close(self):
        if getattr(self, "_halo", None) is not None:
            self._halo.close()
            self._halo = None
        for p in getattr(self, "_opened", []):
            self._lib.sx_ipc_close(p)
        for p in getattr(self, "_own", []):
            self._lib.sx_comm_free(p)
        self._opened, self._own = [], []


class _ThreadShared:
    def __init__(self, world):
        self.world = world
        self.bufs = [None] * world
        self.barrier = threading.Barrier(world)


class ThreadComm(Comm):
    """`world` virtual ranks = threads of one process sharing one GPU (validation only; see module docstring)."""
    graph_safe = False

    def __init__(self, shared, rank):
        self._sh, self.rank, self.world = shared, rank, shared.world

    @staticmethod
    def make(world):
        sh = _ThreadShared(world)
        return [ThreadComm(sh, r) for r in range(world)]

    def _exchange(self, t):
        sh = self._sh
        sh.bufs[self.rank] = t
        if t.is_cuda:
            torch.cuda.synchronize()
        sh.barrier.wait()
        parts = [b.clone() for b in sh.bufs]
        if t.is_cuda:
            torch.cuda.synchronize()
        sh.barrier.wait()                 # everyone has read everyone's buffer → safe to overwrite in place
        return parts

    def all_reduce(self, t):
        parts = self._exchange(t)
        acc = parts[0]
        for q in parts[1:]:
            acc = acc + q                 # same order on every rank → bit-identical results
        t.copy_(acc)
        return t

    def all_gather(self, t):
        return torch.stack(self._exchange(t.contiguous()), dim=0)

    def barrier(self):
        self._sh.barrier.wait()


def run_virtual_ranks(world, fn):
    """Run fn(comm) on `world` ThreadComm ranks; returns the per-rank results (re-raises the first failure)."""
    comms = ThreadComm.make(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001 — surfaced below; abort the barrier so the peers do not hang
            err[r] = e
            comms[r]._sh.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


# ---- batch split of an embarrassingly parallel forward (the ViT over an image's crops) --------------------------------------
def split_batch_forward(fn, x, comm):
    """y = fn(x) with the leading (batch) dimension of x split over the ranks of `comm` and the results all-gathered back in order:
    rank r runs fn on items r, r + world, r + 2·world, … — no communication inside fn. SURVEY.md §8(e) "ViT partitioning": the
    any-res crops of an image are independent (B = 2 … 20, any_res.py:185-189), so in latency mode each GPU encodes its share of
    the crops instead of every GPU encoding all of them. Every rank must call it (collective); fn sees at least one item on every
    rank (a rank without work re-encodes item 0 and its result is dropped), so fn never runs on an empty batch."""
    world, rank = comm.world, comm.rank
    B = x.shape[0]
    if world == 1 or B == 1:
        return fn(x)
    per = -(-B // world)                                   # items per rank, the last ranks may hold padding
    idx = [i for i in range(rank, B, world)]
    pad = per - len(idx)
    sel = torch.as_tensor(idx + [0] * pad, device=x.device)
    y = fn(x.index_select(0, sel)).contiguous()            # [per, ...]
    g = comm.all_gather(y)                                 # [world, per, ...]
    # item i lives at g[i % world, i // world]
    out = g.transpose(0, 1).reshape((per * world,) + tuple(y.shape[1:]))[:B]
    return out.contiguous()


# ---- Megatron sharding of the reference Llama state dict (pure tensor slicing; CPU-testable) ------------------------------
def llama_tp_shard(sd, layer_prefix, rank, tp, nh, hd):
    """Rank `rank`'s slices of one decoder layer. Column-parallel: q/k/v rows of heads [rank·nh/tp, (rank+1)·nh/tp),
    gate/up rows [rank·I/tp, …); row-parallel: the matching COLUMNS of o_proj / down_proj. Σ_r o_r·att_r = o·att."""
    assert nh % tp == 0
    hl = nh // tp * hd
    hs = slice(rank * hl, (rank + 1) * hl)
    p = layer_prefix
    I = sd[p + "mlp.gate_proj.weight"].shape[0]
    assert I % (16 * tp) == 0, "FFN rows per rank must keep the 16-row GLU packing"
    il = I // tp
    isl = slice(rank * il, (rank + 1) * il)
    return {"q": sd[p + "self_attn.q_proj.weight"][hs], "k": sd[p + "self_attn.k_proj.weight"][hs],
            "v": sd[p + "self_attn.v_proj.weight"][hs], "o": sd[p + "self_attn.o_proj.weight"][:, hs],
            "gate": sd[p + "mlp.gate_proj.weight"][isl], "up": sd[p + "mlp.up_proj.weight"][isl],
            "down": sd[p + "mlp.down_proj.weight"][:, isl]}
