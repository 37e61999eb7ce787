"""Temperature-ladder sharding across the GPUs of one node (one process per GPU).

Rank g owns the contiguous rungs [g*Tl, (g+1)*Tl).  The stretch move needs no communication
(complement walkers are drawn inside a rung, red_blue.py:183-193).  Two ways to run the PT sweep
(tempering.py:484-561, 598-649) across ranks:

:class:`LadderPipeline` (default) - neighbour exchange by one-sided puts.  Every rank has a mailbox in
uncached device memory that its two ladder neighbours store into directly (HIP IPC mapping = xGMI peer
stores), followed by a flag the consumer kernel spins on.  ``torch.distributed`` only all-gathers the
IPC handles once; after that ``step(n)`` is one library call per rank.  Philox draws only.  DESIGN 6.1.

:class:`StagedPipeline` - the same protocol and kernels with RCCL point-to-point messages (grouped ncclSend /
ncclRecv) between three host-ordered stages per iteration; for nodes without peer mappings.  DESIGN 6.2.

:class:`ShardedLadder` (automatic fallback; also the teacher-forced path) - RCCL collectives per iteration:

  1. all-gather of the log-likelihoods  [Tl, W] -> [T, W]            RCCL all_gather
  2. EVERY rank replays the whole hot->cold cascade from the gathered ladder (one parallel kernel in
     column form); all ranks reach identical decisions and identical adapted betas
  3. walker rows that change rank travel as [dest id | x | logp] records    RCCL all_to_all
  4. received rows are scattered into free pool slots.

The communication layer of the fallback is ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Its compute engine is injected: the product uses
:class:`HipShardEngine`; the CPU tests drive the same orchestration with a NumPy stand-in.
"""
import numpy as np

try:                                   # torch first: see the process-level note in eryn_amd/_lib.py
    import torch  # noqa: F401
except ImportError:                    # pragma: no cover - the orchestration itself needs torch.distributed
    torch = None


def rung_partition(ntemps, nranks):
    """Contiguous equal shards; returns (rank_of_rung[T], [(begin, end)] per rank)."""
    if ntemps % nranks != 0:
        raise ValueError("ntemps must be a multiple of the number of ranks (T < G: use replicas instead)")
    tl = ntemps // nranks
    bounds = [(g * tl, (g + 1) * tl) for g in range(nranks)]
    return np.repeat(np.arange(nranks, dtype=np.int32), tl), bounds


class _DevArray:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class HipShardEngine:
    """The HIP engine seen through the small interface ShardedLadder needs."""

    def __init__(self, engine, device, share_stream=True):
        import torch
        self.torch = torch
        self.e = engine
        self.device = device
        self.T, self.Tl, self.W, self.D = engine.T, engine.Tl, engine.W, engine.D
        if share_stream:
            # one non-default stream for the kernels and (through torch's stream dependencies) the RCCL
            # collectives; the legacy null stream would serialise against every other stream
            self.stream = torch.cuda.Stream(device)
            torch.cuda.set_stream(self.stream)
            engine.set_stream(self.stream.cuda_stream)
        self.shared_stream = bool(share_stream)
        b = engine.device_buffers()
        self.row_doubles = int(b.row_doubles)
        self.cap = int(b.row_capacity)
        self._gather = torch.as_tensor(_DevArray(b.gather_logl, (self.T, self.W)), device=device)
        self._send = torch.as_tensor(_DevArray(b.send_rows, (self.cap, self.row_doubles)), device=device)
        self._recv = torch.as_tensor(_DevArray(b.recv_rows, (self.cap, self.row_doubles)), device=device)

    def stretch(self, draws=None):
        if draws is None:
            self.e.stretch_iter()
            return None
        keeps = []
        for sp in (0, 1):
            keeps.append(self.e.stretch_split(sp, draws["labels"], draws[f"rint{sp}"], draws[f"u_zz{sp}"],
                                              draws[f"u_acc{sp}"]))
        return keeps

    def local_logl(self):
        b = self.e.device_buffers()                     # the current buffer flips every PT step
        if not self.shared_stream:
            self.e.synchronize()                        # library stream -> visible to the comm stream
        return self.torch.as_tensor(_DevArray(b.logl, (self.Tl, self.W)), device=self.device)

    def gather_buffer(self):
        return self._gather

    def plan(self, rank_of_rung, nranks, rank, draws=None, adapt=True):
        kw = {} if draws is None else dict(iperm=draws["iperm"], i1perm=draws["i1perm"], u_swap=draws["u_swap"])
        # production steps skip the per-iteration D2H of the swap counts (hens_get_counters has the totals)
        send, recv, sel, swaps = self.e.pt_plan_sharded(rank_of_rung, nranks, rank, adapt=adapt,
                                                        want_swaps=draws is not None, **kw)
        return send, recv, sel, swaps

    def send_buffer(self, n):
        return self._send[:n]

    def recv_buffer(self, n):
        return self._recv[:n]

    def finish(self, n_recv):
        self.e.pt_finish_sharded(n_recv)


class ShardedLadder:
    """Drives one rank's shard.  ``dist`` is torch.distributed (already initialised)."""

    def __init__(self, engine, ntemps, dist=None, rank=0, nranks=1, group=None):
        self.eng, self.T = engine, int(ntemps)
        self.dist, self.rank, self.nranks, self.group = dist, int(rank), int(nranks), group
        self.rank_of_rung, self.bounds = rung_partition(self.T, self.nranks)
        self.swaps_accepted = np.zeros(self.T - 1)
        # exercise the collectives even with one rank (lets a 1-GPU box validate the RCCL plumbing)
        self.force_collectives = bool(int(__import__("os").environ.get("HENS_FORCE_COLLECTIVES", "0")))

    def pt_step(self, draws=None, adapt=True):
        """Steps 2-5.  Returns (sel or None, swaps_accepted)."""
        eng, dist = self.eng, self.dist
        local = eng.local_logl()
        full = eng.gather_buffer()
        force = self.force_collectives and dist is not None
        if self.nranks > 1 or force:
            dist.all_gather_into_tensor(full.view(-1), local.reshape(-1), group=self.group)
        else:
            full.copy_(local)
        self._sync_comm()
        send, recv, sel, swaps = eng.plan(self.rank_of_rung, self.nranks, self.rank, draws=draws, adapt=adapt)
        n_send, n_recv = int(send.sum()), int(recv.sum())
        if self.nranks > 1 or force:
            out_buf, in_buf = eng.recv_buffer(n_recv), eng.send_buffer(n_send)
            self._all_to_all(out_buf, in_buf, recv, send)
            self._sync_comm()
        eng.finish(n_recv)
        if swaps is not None:
            self.swaps_accepted = swaps
        return sel, swaps

    def _all_to_all(self, out_buf, in_buf, recv_counts, send_counts):
        """Variable all-to-all of rows.  all_to_all_single where the backend has it (RCCL);
        pairwise isend/irecv otherwise (gloo on CPU)."""
        dist = self.dist
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            dist.all_to_all_single(out_buf, in_buf, output_split_sizes=[int(c) for c in recv_counts],
                                   input_split_sizes=[int(c) for c in send_counts], group=self.group)
            return
        reqs, so, ro = [], 0, 0
        ops = []
        for peer in range(self.nranks):
            ns, nr = int(send_counts[peer]), int(recv_counts[peer])
            if peer != self.rank:
                if ns:
                    ops.append(dist.P2POp(dist.isend, in_buf[so:so + ns].contiguous(), peer, group=self.group))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, out_buf[ro:ro + nr], peer, group=self.group))
            so += ns
            ro += nr
        if ops:
            reqs = dist.batch_isend_irecv(ops)
            for r in reqs:
                r.wait()

    def _sync_comm(self):
        eng = self.eng
        if getattr(eng, "shared_stream", False):
            return                                      # same stream: ordered without a host sync
        if hasattr(eng, "torch") and eng.torch.cuda.is_available():
            eng.torch.cuda.current_stream().synchronize()

    def step(self, n_iters=1, adapt=True):
        """Production (Philox) iterations."""
        for _ in range(int(n_iters)):
            self.eng.stretch()
            if self.T > 1:
                self.pt_step(adapt=adapt)


class LadderPipeline:
    """Sharded stepping by neighbour exchange (include/hipensemble.h, "Ladder pipeline").

    Every rank holds a contiguous rung range and a mailbox its two ladder neighbours store into directly
    (HIP IPC peer mapping: xGMI stores on a multi-GPU node).  ``torch.distributed`` is used ONCE, to
    exchange the 64-byte IPC handles; after that ``step(n)`` is a single library call per rank and the
    ranks synchronise through device-side flags only.  The result is bit-identical to one context
    holding the whole ladder (tests/test_hip_pipeline.py).
    """

    def __init__(self, engine, rank, nranks, dist=None, group=None, device_id=0, selftest=None):
        import os
        if selftest is None:
            selftest = os.environ.get("HENS_PIPE_SELFTEST", "1") != "0"
        self.e = engine
        self.rank, self.nranks = int(rank), int(nranks)
        if self.nranks > 1 and dist is None:
            raise ValueError("nranks > 1 needs torch.distributed to exchange the mailbox handles")
        if self.nranks > 1:
            self._refuse_shared_device(dist, group, device_id)
        if self.nranks > 1 and selftest:
            self._selftest(dist, group, device_id)
        if self.nranks == 1:
            engine.pipe_init(1, 0)
            engine.pipe_connect_local([engine])
            return
        # Failure-atomic across ranks: every rank runs the SAME sequence of collectives whether or not its own
        # pipe_init / pipe_connect worked (e.g. hipIpcOpenMemHandle refused), and all of them raise together - a
        # rank that left early would leave its peers inside a collective that never completes.
        handle, err = None, None
        try:
            handle = engine.pipe_init(self.nranks, self.rank)
        except Exception as exc:                      # noqa: BLE001
            err = f"pipe_init: {exc}"
        got = [None] * self.nranks
        dist.all_gather_object(got, (handle, err), group=group)
        self._raise_if_any(got, "hens_pipe_init")
        try:
            engine.pipe_connect(b"".join(h for h, _ in got))
        except Exception as exc:                      # noqa: BLE001
            err = f"pipe_connect: {exc}"
        got2 = [None] * self.nranks
        dist.all_gather_object(got2, (None, err), group=group)     # also the barrier: nobody stores into a mailbox
        self._raise_if_any(got2, "hens_pipe_connect")               # that is not mapped everywhere yet

    def _refuse_shared_device(self, dist, group, device_id):
        """One rank per GPU.  Ranks that share a device starve each other's flag waits (a resident kernel that spins on a flag
        whose producer cannot get a compute unit never finishes: tests/test_hip_fullsize.py) - allowed only as a dry run of
        the code path: the gloo backend (RCCL itself refuses two ranks on one GPU) or HENS_PIPE_SHARED_GPU=1.  Every rank
        takes part in the collective and all of them raise together."""
        import os
        import socket
        unique = True                 # does `dev` identify the physical GPU?
        try:
            import torch
            props = torch.cuda.get_device_properties(int(device_id))
            dev = getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None)
        except Exception:                             # noqa: BLE001
            dev = None
        if dev is None:                               # ... then from the HIP runtime itself, through the library
            try:
                import ctypes
                from . import _lib
                buf = ctypes.create_string_buffer(64)
                if _lib.load().hens_device_pci_bus_id(int(device_id), buf, 64) == 0 and buf.value:
                    dev = buf.value.decode()
            except Exception:                         # noqa: BLE001
                dev = None
        if dev is None:
            # no physical id at all: the ordinal under the visibility masks (one-rank-per-GPU launchers mask
            # every rank down to "its" GPU, which is then ordinal 0 everywhere) - good enough to warn, not to refuse
            unique = False
            dev = (device_id, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"), os.environ.get("CUDA_VISIBLE_DEVICES"))
        ids = [None] * self.nranks
        dist.all_gather_object(ids, (socket.gethostname(), str(dev), unique), group=group)
        dry_run = dist.get_backend(group) == "gloo" or os.environ.get("HENS_PIPE_SHARED_GPU") == "1"
        if len(set(ids)) < self.nranks and not dry_run and not all(u for _, _, u in ids):
            import warnings
            warnings.warn(f"ladder pipeline: cannot tell whether ranks share a GPU (no device uuid / PCI id from torch): {ids}")
        elif len(set(ids)) < self.nranks and not dry_run:
            raise RuntimeError(f"ladder pipeline: ranks share a GPU ({ids}); one rank per GPU is required outside the dry-run "
                               f"backend (gloo) / HENS_PIPE_SHARED_GPU=1")

    @staticmethod
    def _raise_if_any(results, what):
        bad = [(q, e) for q, (_, e) in enumerate(results) if e]
        if bad:
            raise RuntimeError(f"ladder pipeline: {what} failed on ranks {[q for q, _ in bad]}: {bad[0][1]}")

    def _selftest(self, dist, group, device_id, timeout_s=30.0):
        """hens_pipe_selftest in a throw-away process per rank: a node where peer mappings do not work must show
        up as a failed helper, not as a GPU fault in this process.  Raises RuntimeError if ANY rank failed."""
        import os
        import shutil
        import subprocess
        import sys
        import tempfile
        box = [tempfile.mkdtemp(prefix="hens_probe_") if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            r = subprocess.run([sys.executable, "-m", "eryn_amd.pipe_probe", str(int(device_id)), str(self.rank),
                                str(self.nranks), box[0], str(timeout_s)], cwd=root, capture_output=True, text=True,
                               timeout=3 * timeout_s + 60)
            ok, msg = r.returncode == 0, (r.stdout + r.stderr).strip()[-300:]
        except subprocess.TimeoutExpired:
            ok, msg = False, "helper process timed out"
        results = [None] * self.nranks
        dist.all_gather_object(results, (ok, msg), group=group)
        if self.rank == 0:
            shutil.rmtree(box[0], ignore_errors=True)
        bad = [(q, m) for q, (o, m) in enumerate(results) if not o]
        if bad:
            raise RuntimeError(f"ladder pipeline self-test failed on ranks {[q for q, _ in bad]}: {bad[0][1]}")

    @staticmethod
    def connect_local(engines):
        """Several shards in ONE process (tests): engines in rank order."""
        for r, e in enumerate(engines):
            e.pipe_init(len(engines), r)
        for e in engines:
            e.pipe_connect_local(engines)

    def step(self, n_iters=1):
        self.e.step(n_iters)


class StagedPipeline:
    """The ladder pipeline's protocol with RCCL point-to-point messages (include/hipensemble.h, "Staged transport").

    Same kernels and message contents as :class:`LadderPipeline`; the stores go into local outboxes and
    ``torch.distributed`` isend / irecv (grouped ncclSend / ncclRecv over xGMI on ROCm) carries them between the
    stages, ordered by the host.  RCCL cannot pull, so the boundary rung's rows travel with the LDN message and the
    rows that move down travel as a dense [W, D] block.  It exists for nodes without peer mappings; the one-sided
    transport is faster (no host ordering, sparse rows, one library call per n iterations).
    """

    def __init__(self, engine, rank, nranks, dist, device, group=None):
        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.e, self.rank, self.nranks, self.device = engine, int(rank), int(nranks), device
        self.top, self.bot = self.rank + 1 < self.nranks, self.rank > 0
        self.host_staged = dist.get_backend(group) == "gloo"       # gloo moves CPU tensors only (tests on one GPU)
        # one stream for the kernels and the collectives: ordered without host synchronisation
        self.stream = torch.cuda.Stream(device)
        torch.cuda.set_stream(self.stream)
        engine.set_stream(self.stream.cuda_stream)
        engine.pipe_init(self.nranks, self.rank)
        engine.pipe_connect_staged()

    def _t(self, ptr, n, typestr="<f8"):
        return self.torch.as_tensor(_DevArray(ptr, (int(n),), typestr), device=self.device)

    def _exchange(self, sends, recvs):
        """sends / recvs: [(tensor, peer)].  One grouped batch of point-to-point operations."""
        dist, torch = self.dist, self.torch
        if not sends and not recvs:
            return
        if self.host_staged:
            torch.cuda.current_stream().synchronize()
            bufs = [torch.empty(t.shape, dtype=t.dtype) for t, _ in recvs]
            ops = [dist.P2POp(dist.isend, t.cpu(), p, group=self.group) for t, p in sends]
            ops += [dist.P2POp(dist.irecv, b, p, group=self.group) for b, (_, p) in zip(bufs, recvs)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for b, (t, _) in zip(bufs, recvs):
                t.copy_(b)
            return
        ops = [dist.P2POp(dist.isend, t, p, group=self.group) for t, p in sends]
        ops += [dist.P2POp(dist.irecv, t, p, group=self.group) for t, p in recvs]
        for r in dist.batch_isend_irecv(ops):
            r.wait()

    def step(self, n_iters=1):
        e, up, dn = self.e, self.rank + 1, self.rank - 1
        for _ in range(int(n_iters)):
            r = e.pipe_regions()
            lp, rows, nc = r.lp_doubles, r.row_doubles, r.cnt_words
            e.pipe_stage(0)                                                        # move; publishes the boundary rung
            sends = [(self._t(r.ldn_out, lp), up), (self._t(r.ldn_rows_out, rows), up)] if self.top else []
            recvs = [(self._t(r.ldn_in, lp), dn), (self._t(r.ldn_rows_in, rows), dn)] if self.bot else []
            self._exchange(sends, recvs)                                           # LDN: cold -> hot
            if self.top:
                self._exchange([], [(self._t(r.lup_in, lp), up)])                  # LUP: hot -> cold
            e.pipe_stage(1)                                                        # walk
            if self.bot:
                self._exchange([(self._t(r.lup_out, lp), dn)], [])
            if self.top:                                                           # ROWS: hot -> cold.  Before my bottom
                self._exchange([], [(self._t(r.rows_in, rows), up)])               # kernel: a walker may fall through all
            e.pipe_stage(2)                                                        # my rungs in one sweep
            if self.bot:
                self._exchange([(self._t(r.rows_out, rows), dn)], [])
            cnt = self._t(r.cnt_out, nc, "<i4")                                    # CNT: every pair's owner -> all
            if self.nranks > 1:
                if self.host_staged:
                    c = cnt.cpu()
                    self.dist.all_reduce(c, group=self.group)
                    cnt.copy_(c)
                else:
                    self.dist.all_reduce(cnt, group=self.group)
            self._t(r.cnt_in, nc, "<i4").copy_(cnt)


class RcclPipeline:
    """The staged protocol with the messages sent by the LIBRARY (``hens_comm_init``): grouped ncclSend / ncclRecv between ladder
    neighbours and one all-reduce of the swap counts per sweep, enqueued between the stage launches on the context's stream -
    ``step(n)`` is one C call for n iterations, no Python in the loop (:class:`StagedPipeline` orders the same messages from the
    host through ``torch.distributed`` and stays for the CPU tests' gloo backend).  ``dist`` is used once: to hand rank 0's
    ncclUniqueId to the other ranks."""

    def __init__(self, engine, rank, nranks, dist=None, group=None):
        self.e, self.rank, self.nranks = engine, int(rank), int(nranks)
        uid = [engine.comm_unique_id() if self.rank == 0 else None]
        if self.nranks > 1:
            if dist is None:
                raise ValueError("nranks > 1 needs a torch.distributed group to share the communicator's id")
            dist.broadcast_object_list(uid, src=0, group=group)
        engine.comm_init(self.nranks, self.rank, uid[0])

    def step(self, n_iters=1):
        self.e.step(int(n_iters))

    def synchronize(self):
        self.e.synchronize()

    def close(self):
        self.e.comm_destroy()
