"""Communication of the row-sharded solver (ipsolver/sharded.py): the collectives over
``torch.distributed`` (backend "nccl" = RCCL over xGMI; gloo in the CPU tests) and the peer
mailboxes -- hipIpc-mapped device memory the loop's kernels write into directly
(csrc/peer.hip, csrc/resident.hip)."""
import ctypes

import torch
import torch.distributed as dist


class ShardComm:
    """The collectives of the sharded solver over ``torch.distributed``."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.backend = dist.get_backend(group) if self.on else "none"
        # (ipc_*: batches / iterations of the device loop that ran on the peer mailboxes --
        # none of the four counters above moves between their boundaries)
        self.stats = {"all_reduce": 0, "all_reduce_bytes": 0, "exchange": 0, "exchange_bytes": 0,
                      "ipc_batches": 0, "ipc_iterations": 0, "ipc_gathers": 0, "ipc_exchanges": 0}
        # the group's peer mailboxes once they are mapped (Sharding.mailbox): the few-scalar
        # collectives and the halo exchanges of the OUTER loops then go through them too --
        # one kernel each, no call into torch.distributed (over gloo, the tests' and the
        # one-GPU rehearsal's backend, such a call is 100-250 us of host staging; over RCCL
        # ~25 us + a synchronisation)
        self.mbox = None
        self.on_mailbox_drop = None

    def all_reduce(self, t, op="sum"):
        """In place on a torch tensor (CUDA under nccl; CUDA tensors are staged through the
        host under gloo, a test-only combination)."""
        if self.world == 1:
            return
        self.stats["all_reduce"] += 1
        self.stats["all_reduce_bytes"] += t.numel() * t.element_size()
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        if t.is_cuda and self.backend != "nccl":
            h = t.cpu()
            dist.all_reduce(h, op=rop, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=rop, group=self.group)

    def _gather_floats(self, values):
        """Every rank's values through the mailboxes: a list (by rank) of lists, or None when
        this collective cannot take that road (no mailbox, too many values)."""
        mb = self.mbox
        if mb is None or not 1 <= len(values) <= mb.NQ:
            return None
        self.stats["ipc_gathers"] += 1
        try:
            return mb.allgather(values)
        except MailboxOutOfStep as exc:
            # this collective's own wait timed out -- on every rank (a rank whose sequence
            # numbers ran ahead waits for tags nobody sends, the others for its): the group
            # leaves the mailboxes together and repeats the collective on torch.distributed
            from warnings import warn
            warn("row-sharded solver: %s -- falling back to the torch.distributed transport for "
                 "the rest of this process" % exc)
            self.mbox = None
            if self.on_mailbox_drop is not None:
                self.on_mailbox_drop()
            return None

    def reduce_floats(self, values, op="sum", device=None):
        """All-reduce of a few host scalars (one blocking round)."""
        if self.world == 1:
            return [float(v) for v in values]
        parts = self._gather_floats(values)
        if parts is not None:                 # (combined in rank order: the same bits everywhere)
            comb = {"sum": sum, "max": max, "min": min}[op]
            return [comb(p[q] for p in parts) for q in range(len(values))]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.to(device if device is not None
                     else torch.device("cuda", torch.cuda.current_device()))
        self.all_reduce(t, op)
        return t.tolist()

    def reduce_mixed(self, sums=(), maxs=(), mins=()):
        """Sums, maxima and minima of host scalars over the ranks in ONE collective (an
        all-gather of every rank's values, combined locally in rank order: bit-identical on
        every rank)."""
        ns, nx, nn = len(sums), len(maxs), len(mins)
        if self.world == 1:
            return [float(v) for v in sums], [float(v) for v in maxs], [float(v) for v in mins]
        parts = self._gather_floats([*sums, *maxs, *mins])
        if parts is not None:
            return ([sum(p[q] for p in parts) for q in range(ns)],
                    [max(p[q] for p in parts) for q in range(ns, ns + nx)],
                    [min(p[q] for p in parts) for q in range(ns + nx, ns + nx + nn)])
        t = torch.tensor([float(v) for v in (*sums, *maxs, *mins)], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.stats["all_reduce"] += 1
        self.stats["all_reduce_bytes"] += 8 * t.numel() * self.world
        dist.all_gather(parts, t, group=self.group)
        g = torch.stack(parts).cpu()
        return (g[:, :ns].sum(0).tolist(), g[:, ns:ns + nx].max(0).values.tolist() if nx else [],
                g[:, ns + nx:].min(0).values.tolist() if nn else [])

    def _ipc_exchange(self, whole, jobs):
        """The halo update of ``jobs`` (segments of the ONE local tensor ``whole``) through the
        mailboxes; False when it cannot take that road."""
        mb = self.mbox
        if mb is None or whole is None or not whole.is_cuda or not 1 <= len(jobs) <= 4:
            return False
        if not mb.exchange(whole, jobs):
            return False
        self.stats["ipc_exchanges"] += 1
        return True

    def exchange_many(self, jobs, whole=None):
        """Several halo updates (tensor, own_lo, own_hi, send_left, send_right) as ONE batch of
        point-to-point operations (the segments of a stacked vector; ``whole``: the local
        tensor they are slices of, in order)."""
        if self.world == 1:
            return
        if self._ipc_exchange(whole, jobs):
            return
        ops, staged, r = [], [], self.rank
        for t, own_lo, own_hi, send_left, send_right in jobs:
            n = t.numel()
            stage = t.is_cuda and self.backend != "nccl"
            buf = t.cpu() if stage else t
            if stage:
                staged.append((t, buf, own_lo, own_hi))
            if r > 0:
                if send_left:
                    ops.append(dist.P2POp(dist.isend, buf[own_lo:own_lo + send_left], r - 1,
                                          self.group))
                if own_lo:
                    ops.append(dist.P2POp(dist.irecv, buf[0:own_lo], r - 1, self.group))
            if r < self.world - 1:
                if send_right:
                    ops.append(dist.P2POp(dist.isend, buf[own_hi - send_right:own_hi], r + 1,
                                          self.group))
                if n - own_hi:
                    ops.append(dist.P2POp(dist.irecv, buf[own_hi:n], r + 1, self.group))
            self.stats["exchange_bytes"] += 8 * (send_left * (r > 0)
                                                 + send_right * (r < self.world - 1))
        if ops:
            self.stats["exchange"] += 1
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, buf, own_lo, own_hi in staged:
            if own_lo:
                t[0:own_lo].copy_(buf[0:own_lo])
            if t.numel() - own_hi:
                t[own_hi:].copy_(buf[own_hi:])

    def prepare_exchange(self, t, own_lo, own_hi, send_left, send_right):
        """``prepare_exchange_many`` for one buffer."""
        return self.prepare_exchange_many([(t, own_lo, own_hi, send_left, send_right)])

    def prepare_exchange_many(self, jobs):
        """The halo updates of ``exchange_many`` for FIXED buffers as a callable: the
        point-to-point operations are built once (the device-resident loop repeats the same
        exchange every iteration; building them costs more host time than issuing them) and
        issued as one batch.  CUDA buffers under gloo (a test-only combination) fall back to
        the staged form."""
        if self.world == 1:
            return lambda: None
        jobs = list(jobs)
        if any(j[0].is_cuda for j in jobs) and self.backend != "nccl":
            return lambda: self.exchange_many(jobs)
        r, ops, nbytes = self.rank, [], 0
        for t, own_lo, own_hi, send_left, send_right in jobs:
            n = t.numel()
            if r > 0:
                if send_left:
                    ops.append(dist.P2POp(dist.isend, t[own_lo:own_lo + send_left], r - 1,
                                          self.group))
                if own_lo:
                    ops.append(dist.P2POp(dist.irecv, t[0:own_lo], r - 1, self.group))
            if r < self.world - 1:
                if send_right:
                    ops.append(dist.P2POp(dist.isend, t[own_hi - send_right:own_hi], r + 1,
                                          self.group))
                if n - own_hi:
                    ops.append(dist.P2POp(dist.irecv, t[own_hi:n], r + 1, self.group))
            nbytes += 8 * (send_left * (r > 0) + send_right * (r < self.world - 1))
        stats, batch = self.stats, dist.batch_isend_irecv

        def go():
            if ops:
                stats["exchange"] += 1
                stats["exchange_bytes"] += nbytes
                for w in batch(ops):
                    w.wait()
        return go

    def exchange(self, t, own_lo, own_hi, send_left, send_right):
        """Halo update of the local extended 1-D tensor ``t``: entries [0, own_lo) come from
        the left neighbour's last own entries, [own_hi, len) from the right neighbour's
        first; this rank sends its first ``send_left`` / last ``send_right`` own entries."""
        if self.world == 1:
            return
        if self._ipc_exchange(t, [(t, own_lo, own_hi, send_left, send_right)]):
            return
        n = t.numel()
        stage = t.is_cuda and self.backend != "nccl"
        buf = t.cpu() if stage else t
        ops, r = [], self.rank
        if r > 0:
            if send_left:
                ops.append(dist.P2POp(dist.isend, buf[own_lo:own_lo + send_left].contiguous()
                                      if stage else buf[own_lo:own_lo + send_left], r - 1,
                                      self.group))
            if own_lo:
                ops.append(dist.P2POp(dist.irecv, buf[0:own_lo], r - 1, self.group))
        if r < self.world - 1:
            if send_right:
                ops.append(dist.P2POp(dist.isend, buf[own_hi - send_right:own_hi], r + 1,
                                      self.group))
            if n - own_hi:
                ops.append(dist.P2POp(dist.irecv, buf[own_hi:n], r + 1, self.group))
        if ops:
            self.stats["exchange"] += 1
            self.stats["exchange_bytes"] += 8 * (send_left * (r > 0)
                                                 + send_right * (r < self.world - 1))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if stage:
            if own_lo:
                t[0:own_lo].copy_(buf[0:own_lo])
            if n - own_hi:
                t[own_hi:n].copy_(buf[own_hi:n])


class MailboxOutOfStep(RuntimeError):
    """The wait of a mailbox all-gather timed out (nothing but this collective is lost)."""


class PeerMailbox:
    """This rank's mailbox and its peers', mapped through hipIpc (csrc/peer.hip): the transport
    of the device-resident loop's scalars and halo when every rank of the group runs on this
    node.  Construction is collective (the handles travel through
    ``torch.distributed.all_gather_object`` once); afterwards the mailboxes are touched by
    kernels only.  ``ok`` is False -- on EVERY rank -- when any rank could not map a peer
    (ranks on different nodes, IPC refused): the loop then stays on ``torch.distributed``."""

    def __init__(self, comm, halo_cap):
        from . import _hip
        self._hip, self.comm = _hip, comm
        lib = self.lib = _hip.load()
        self.handle, self.ok, self.error = None, False, None
        world, rank = comm.world, comm.rank
        blob, cap = None, int(halo_cap)
        try:
            caps = [None] * world
            dist.all_gather_object(caps, cap, group=comm.group)
            cap = max(caps)
            self.handle = lib.ipx_peer_create(rank, world, cap)
            if not self.handle:
                raise _hip.IpxError("ipx_peer_create failed: " + lib.ipx_last_error().decode())
            buf = ctypes.create_string_buffer(lib.ipx_peer_handle_bytes())
            _hip.call("ipx_peer_export", ctypes.c_void_p(self.handle), buf)
            blob = (_host_id(), buf.raw)
        except Exception as exc:                 # keep going: the group decides together below
            self.error = repr(exc)
        blobs = [None] * world
        dist.all_gather_object(blobs, blob, group=comm.group)
        good = all(b is not None and b[0] == blobs[0][0] for b in blobs)
        if good:
            try:
                for r, b in enumerate(blobs):
                    if r != rank:
                        _hip.call("ipx_peer_import", ctypes.c_void_p(self.handle), r, b[1])
            except Exception as exc:
                self.error, good = repr(exc), False
        elif self.error is None:
            self.error = "ranks on different hosts, or a peer could not export its mailbox"
        flags = [None] * world
        dist.all_gather_object(flags, bool(good), group=comm.group)
        self.ok = all(flags)
        if self.ok:
            # one all-reduce through the mailboxes against the known answer, on every rank; a
            # group in which it fails anywhere (stores that do not arrive, a wait that times
            # out) falls back to torch.distributed TOGETHER instead of one rank raising
            try:
                self.check()
                passed = True
            except Exception as exc:
                self.error, passed = repr(exc), False
            dist.all_gather_object(flags, passed, group=comm.group)
            self.ok = all(flags)
            if not self.ok and self.error is None:
                self.error = "the mailbox self-test failed on another rank"
        if not self.ok:
            self.close()

    def check(self):
        """One all-reduce through the mailboxes against the known answer."""
        out = self.allreduce([float(self.comm.rank + 1), 1.0])
        w = self.comm.world
        if out != [w * (w + 1) / 2.0, float(w)]:
            raise self._hip.IpxError("peer mailbox self-test failed: %r" % (out,))

    def allreduce(self, values, reps=1):
        """Sum of up to 8 host scalars over the ranks through the mailboxes (set-up checks and
        bench.py's latency probe; the loop's reductions never pass through the host)."""
        from . import device as dv
        dev = dv.ctx().device
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        out = torch.zeros_like(t)
        failed = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hip.call("ipx_peer_allreduce", ctypes.c_void_p(self.handle), t.numel(),
                       ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                       ctypes.c_void_p(failed.data_ptr()), int(reps), dv.stream_ptr())
        if int(failed.item()):
            raise self._hip.IpxError("peer mailbox: a wait for a peer timed out")
        return out.tolist()

    NQ = 7            # IPX_PEER_NQ - 1: scalars per rank of one all-gather (+ the failure word)

    def _buffers(self):
        if getattr(self, "_gout", None) is None:
            from . import device as dv
            dev = dv.ctx().device
            self._gout = torch.zeros(self.comm.world * self.NQ + 2, dtype=torch.float64, device=dev)
            self._failed = torch.zeros(2, dtype=torch.int32, device=dev)
            self._vals = (ctypes.c_double * self.NQ)()
            self._geom = (ctypes.c_int64 * 24)()
        return self._gout, self._failed

    def allgather(self, values):
        """Every rank's ``values`` (<= 8 host scalars) as a list by rank: one kernel that takes
        them by value, one blocking read (csrc/peer.hip ipx_peer_allgather).  Raises when a
        wait of this or of an earlier mailbox collective (a halo exchange) timed out."""
        from . import device as dv
        out, failed = self._buffers()
        nq, w = len(values), self.comm.world
        for q, v in enumerate(values):
            self._vals[q] = float(v)
        self._hip.call("ipx_peer_allgather", ctypes.c_void_p(self.handle), nq, self._vals,
                       ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(failed.data_ptr()),
                       dv.stream_ptr())
        got = dv.read_doubles(out, w * nq + 2)
        if got[w * nq] != 0.0:
            raise self._hip.IpxError("peer mailbox: a halo exchange timed out on a rank of the "
                                     "group (a rank died or fell out of step); the halos it "
                                     "left cannot be trusted")
        if got[w * nq + 1] != 0.0:
            raise MailboxOutOfStep("peer mailbox: a wait for a peer timed out (a rank of the "
                                   "group died or fell out of step)")
        return [got[r * nq:(r + 1) * nq] for r in range(w)]

    def exchange(self, whole, jobs):
        """Halo update of the segments ``jobs`` = (slice of ``whole``, own_lo, own_hi, send_left,
        send_right) of one local CUDA tensor: one kernel, no synchronisation (a wait that times
        out is reported by the next ``allgather``).  False: not representable (a slice that is
        not a view into ``whole``)."""
        from . import device as dv
        _, failed = self._buffers()
        base, g = whole.data_ptr(), self._geom
        for k, (t, own_lo, own_hi, send_left, send_right) in enumerate(jobs):
            off = (t.data_ptr() - base) // 8
            if off < 0 or off + t.numel() > whole.numel() or not t.is_contiguous():
                return False
            g[6 * k:6 * k + 6] = [off, off + own_lo, off + own_hi, off + t.numel(),
                                  send_left if self.comm.rank > 0 else 0,
                                  send_right if self.comm.rank < self.comm.world - 1 else 0]
        self._hip.call("ipx_peer_exchange", ctypes.c_void_p(self.handle),
                       ctypes.c_void_p(base), len(jobs), g, ctypes.c_void_p(failed.data_ptr()),
                       dv.stream_ptr())
        return True

    def attach_resident(self):
        """The hand-off buffers of the resident loop kernel's PEER form (csrc/resident.hip), sized
        for the largest launch the kernel admits; collective, once per mailbox.  True when every
        rank mapped every other rank's (else the group keeps the separate launches)."""
        if getattr(self, "_resident", None) is None:
            from . import cg_fused
            _hip, lib, comm = self._hip, self.lib, self.comm
            lim = cg_fused.resident_limits()
            words = int(lib.ipx_cg_resident_ll_words(lim["max_wg"], lim["halo"]))
            blob = None
            try:
                _hip.call("ipx_peer_attach_resident", ctypes.c_void_p(self.handle), words)
                buf = ctypes.create_string_buffer(lib.ipx_peer_handle_bytes())
                _hip.call("ipx_peer_export_resident", ctypes.c_void_p(self.handle), buf)
                blob = buf.raw
            except Exception as exc:
                self.resident_error = repr(exc)
            blobs = [None] * comm.world
            dist.all_gather_object(blobs, blob, group=comm.group)
            good = all(b is not None for b in blobs)
            if good:
                try:
                    for r, b in enumerate(blobs):
                        if r != comm.rank:
                            _hip.call("ipx_peer_import_resident", ctypes.c_void_p(self.handle), r, b)
                    good = bool(lib.ipx_peer_resident_ready(ctypes.c_void_p(self.handle)))
                except Exception as exc:
                    self.resident_error, good = repr(exc), False
            flags = [None] * comm.world
            dist.all_gather_object(flags, bool(good), group=comm.group)
            self._resident = all(flags)
        return self._resident

    def resident_launches(self):
        return int(self.lib.ipx_peer_resident_launches(ctypes.c_void_p(self.handle)))

    def pingpong(self, reps=200):
        """Round-trip time (us) of one tagged word between this rank and each neighbour, measured
        inside one kernel per pair (csrc/peer.hip k_peer_pingpong): what a cross-GPU hand-off
        costs on this node.  Collective (two rounds: pairs (0,1)(2,3).., then (1,2)(3,4)..);
        returns {neighbour rank: us per round trip} for this rank's neighbours."""
        from . import device as dv
        comm, out = self.comm, {}
        ticks = torch.zeros(2, dtype=torch.int64, device=dv.ctx().device)
        for parity in (0, 1):
            r = comm.rank
            partner = r + 1 if (r - parity) % 2 == 0 else r - 1
            if r < parity or partner < 0 or partner >= comm.world:
                partner = -1
            self._hip.call("ipx_peer_pingpong", ctypes.c_void_p(self.handle), int(partner), int(reps),
                           ctypes.c_void_p(ticks.data_ptr()), dv.stream_ptr())
            if partner >= 0:
                t = ticks.tolist()
                if t[1]:
                    raise self._hip.IpxError("peer mailbox: the ping-pong with rank %d timed out"
                                             % partner)
                out[partner] = t[0] / 100.0 / reps          # 100 MHz ticks -> us per round trip
        return out

    def set_timeout(self, seconds):
        """Deadline of a kernel's wait for a peer's word (default 10 s; past it: stop code 7,
        the group falls back to torch.distributed together)."""
        self._hip.call("ipx_peer_set_timeout", ctypes.c_void_p(self.handle), float(seconds))

    def sequence(self):
        out = (ctypes.c_int64 * 2)()
        self._hip.call("ipx_peer_sequence", ctypes.c_void_p(self.handle), out)
        return int(out[0]), int(out[1])

    def fused_launches(self):
        """Loop kernels so far that did their part of a collective in their own prologue
        (``ipx_shard2_ext.fuse_comm``; 2 per iteration: 3 launches instead of 5)."""
        return int(self.lib.ipx_peer_fused_launches(ctypes.c_void_p(self.handle)))

    def close(self):
        h, self.handle = self.handle, None
        if h:
            try:
                torch.cuda.synchronize()
                self.lib.ipx_peer_destroy(ctypes.c_void_p(h))
            except Exception:
                pass

    __del__ = close


def _device_id():
    """What tells two ranks that they run on the same GPU: host + the device's UUID (its PCI
    address where torch does not expose one)."""
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = getattr(props, "uuid", None)
    if ident is None:
        ident = (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1),
                 getattr(props, "pci_device_id", torch.cuda.current_device()))
    return _host_id() + ":" + str(ident)


def _host_id():
    import socket
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            return socket.gethostname() + ":" + f.read().strip()
    except OSError:
        return socket.gethostname()


