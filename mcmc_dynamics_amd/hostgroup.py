"""Host-side process group for one-process-per-GPU runs, without PyTorch.

The data path of a multi-GPU evaluation is ONE RCCL all-reduce inside ``libmcd_hip.so``; what the host processes
need from each other is tiny and rare: rank 0's 128-byte RCCL unique id at start-up, a barrier and a max-over-ranks
around a timed region, identical walker start positions and sampler seeds (``Runner.__call__``), an occasional checksum.
This module does that over plain TCP sockets in a star around rank 0, so that the package imports no PyTorch and
every process keeps the ROCm libraries it was built against (``torch`` would map its own bundled HIP runtime and RCCL
into the process first).  The reference has no counterpart: its only parallelism is a pathos process pool over
walkers (analysis/runner.py:398-403).

Rendezvous (how the other ranks find rank 0's listening socket), from the launcher's environment
(``RANK``, ``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT`` as ``python -m torch.distributed.run`` / ``torchrun`` set them):

* default, single node -- rank 0 binds an ephemeral port on 127.0.0.1 and publishes it in a small file under the
  temporary directory, keyed by (MASTER_PORT, launcher pid, restart count).  ``MASTER_PORT`` itself is NOT used as a
  listening port: the torchrun agent's own store already owns it.
* ``MCD_RDZV_PORT=<port>`` (multi-node, or launchers whose workers have different parents) -- rank 0 listens on that
  port on all interfaces, the others connect to ``MASTER_ADDR:<port>``.

Collectives are deterministic: the hub combines contributions in rank order.

Abort channel (round 3).  The collectives above are request / response on one socket per rank and a rank that sits in a
library call cannot answer them.  So that a rank which FAILS (an exception inside a block of ``mcd_stretch_move``, a refused
launch) can tell its peers -- who would otherwise wait inside the device all-reduce until the library's collective deadline
-- every rank keeps a second connection to the hub with a daemon thread reading it: ``abort(reason)`` sends one message,
the hub relays it to everybody, and every rank's callbacks (``on_abort``: ``Context.abort``, i.e. ``mcd_ctx_abort``) run on
the listener thread within milliseconds.  After an abort every collective of the group raises ``HostGroupError``.
"""
import json
import os
import select
import socket
import struct
import tempfile
import threading
import time
import zlib

import numpy as np

_DTYPES = {"f8": np.float64, "i8": np.int64}


class HostGroupError(RuntimeError):
    """Rendezvous failed, a peer vanished, or a collective timed out."""


def _send(sock, header, payload=b""):
    h = json.dumps(header).encode()
    sock.sendall(struct.pack("<II", len(h), len(payload)) + h + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise HostGroupError("peer closed the connection")
        buf.extend(chunk)
    return bytes(buf)


def _recv(sock):
    hl, pl = struct.unpack("<II", _recv_exact(sock, 8))
    header = json.loads(_recv_exact(sock, hl).decode())
    return header, (_recv_exact(sock, pl) if pl else b"")


def rendezvous_file(env=None):
    env = os.environ if env is None else env
    explicit = env.get("MCD_RDZV_FILE")
    if explicit:
        return explicit
    # all workers of one torchrun agent share the agent as parent; other launchers: fall back to the port alone
    parent = os.getppid() if env.get("TORCHELASTIC_RUN_ID") is not None else 0
    name = "mcd_rdzv_{0}_{1}_{2}_{3}_{4}.json".format(os.getuid(), env.get("MASTER_PORT", "0"), parent,
                                                     env.get("TORCHELASTIC_RUN_ID", "none"),
                                                     env.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(tempfile.gettempdir(), name)


class HostGroup(object):
    """Star-topology group: rank 0 is the hub.  ``timeout`` (seconds) bounds the rendezvous and every collective."""

    def __init__(self, rank, world, timeout=300.0, env=None):
        self.rank, self.world = int(rank), int(world)
        self.timeout = float(timeout)
        self._peers = {}           # hub: rank -> socket
        self._hub = None           # others: socket to rank 0
        self._listener = None
        self._file = None
        self._seq = 0
        self._ctl_peers = {}       # hub: rank -> control socket
        self._ctl_hub = None       # others: control socket to rank 0
        self._ctl_thread = None
        self._ctl_lock = threading.Lock()
        self._abort_callbacks = []
        self.aborted = None        # reason (str) once any rank has called abort()
        self._closing = False
        if not 0 <= self.rank < self.world:
            raise HostGroupError("rank {0} outside world of size {1}".format(rank, world))
        if self.world > 1:
            self._connect(os.environ if env is None else env)
            self._ctl_thread = threading.Thread(target=self._control_loop, name="mcd-hostgroup-control", daemon=True)
            self._ctl_thread.start()
            # a normal interpreter exit says goodbye on the control connection; a process that dies does not, and its
            # peers take that as a failure of the job (see _control_loop)
            import atexit
            import weakref
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())

    @classmethod
    def from_env(cls, timeout=300.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)

    # ------------------------------------------------------------------ rendezvous
    def _connect(self, env):
        port = env.get("MCD_RDZV_PORT")
        token = "{0}-{1}".format(env.get("TORCHELASTIC_RUN_ID", "none"), env.get("MASTER_PORT", "0"))
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            if port:
                srv.bind(("", int(port)))
            else:
                srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            self._listener = srv
            if not port:
                self._file = rendezvous_file(env)
                tmp = "{0}.{1}.tmp".format(self._file, os.getpid())
                with open(tmp, "w") as f:
                    json.dump({"port": srv.getsockname()[1], "pid": os.getpid(), "token": token, "world": self.world}, f)
                os.replace(tmp, self._file)                       # atomic: readers see the old or the new file
            while len(self._peers) < self.world - 1 or len(self._ctl_peers) < self.world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise HostGroupError("rendezvous: {0} of {1} ranks connected within {2:.0f} s".format(
                        len(self._peers) + 1, self.world, self.timeout))
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    hello, _ = _recv(conn)
                except (HostGroupError, socket.timeout, ValueError):
                    conn.close()
                    continue
                r = hello.get("rank")
                table = self._ctl_peers if hello.get("control") else self._peers
                if hello.get("token") != token or hello.get("world") != self.world or not isinstance(r, int) \
                        or not 0 < r < self.world or r in table:
                    conn.close()                                   # a stray or stale client: ignore it
                    continue
                _send(conn, {"ok": True})
                if hello.get("control"):
                    conn.settimeout(None)
                table[r] = conn
        else:
            addr = env.get("MASTER_ADDR", "127.0.0.1") if port else "127.0.0.1"
            last = "no rendezvous file yet"
            while True:
                if time.monotonic() > deadline:
                    raise HostGroupError("rendezvous: rank {0} could not reach rank 0 within {1:.0f} s ({2})".format(
                        self.rank, self.timeout, last))
                try:
                    if port:
                        target = int(port)
                    else:
                        with open(rendezvous_file(env)) as f:
                            info = json.load(f)
                        if info.get("token") != token or info.get("world") != self.world:
                            raise ValueError("rendezvous file belongs to another job")
                        target = int(info["port"])
                    socks = []
                    for control in (False, True):
                        s = socket.create_connection((addr, target), timeout=5.0)
                        socks.append(s)
                        s.settimeout(self.timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        _send(s, {"rank": self.rank, "world": self.world, "token": token, "control": control})
                        ack, _ = _recv(s)
                        if not ack.get("ok"):
                            for x in socks:
                                x.close()
                            raise ValueError("hub refused the connection")
                    self._hub, self._ctl_hub = socks
                    self._ctl_hub.settimeout(None)
                    return
                except (OSError, ValueError, KeyError, HostGroupError) as exc:   # stale file, hub not up yet: retry
                    last = repr(exc)
                    time.sleep(0.05)

    # ------------------------------------------------------------------ abort channel
    def on_abort(self, callback):
        """``callback(reason)`` runs on the listener thread of THIS rank when any rank calls ``abort`` (its own included)."""
        with self._ctl_lock:
            self._abort_callbacks.append(callback)
            reason = self.aborted
        if reason is not None:
            callback(reason)

    def abort(self, reason):
        """Tell every rank of the group that this rank failed: their ``on_abort`` callbacks run (``mcd_ctx_abort``: a peer
        waiting inside the device all-reduce returns an error at once) and every later collective of the group raises.
        Never raises itself: it is called from exception handlers."""
        reason = "rank {0}: {1}".format(self.rank, str(reason)[:500])
        self._deliver_abort(reason)
        try:
            if self.world > 1:
                if self.rank == 0:
                    self._relay_abort(reason, skip=None)
                elif self._ctl_hub is not None:
                    _send(self._ctl_hub, {"kind": "abort", "reason": reason})
        except (OSError, HostGroupError):
            pass

    def _deliver_abort(self, reason):
        with self._ctl_lock:
            if self.aborted is not None:
                return
            self.aborted = reason
            callbacks = list(self._abort_callbacks)
        for cb in callbacks:
            try:
                cb(reason)
            except Exception:
                pass

    def _relay_abort(self, reason, skip):
        for r, sock in list(self._ctl_peers.items()):
            if r == skip:
                continue
            try:
                _send(sock, {"kind": "abort", "reason": reason})
            except (OSError, HostGroupError):
                pass

    def _control_loop(self):
        """Daemon thread: wait for abort messages on the control connection(s)."""
        try:
            while not self._closing:
                socks = list(self._ctl_peers.values()) if self.rank == 0 else [self._ctl_hub]
                socks = [s for s in socks if s is not None]
                if not socks:
                    return
                ready, _, _ = select.select(socks, [], [], 0.2)
                for sock in ready:
                    try:
                        header, _ = _recv(sock)
                    except (HostGroupError, OSError, ValueError, struct.error):
                        # the peer is gone.  Without a goodbye (close() sends one) it died: that is a failure of the job
                        if not self._closing:
                            who = next((r for r, s in self._ctl_peers.items() if s is sock), 0)
                            reason = "rank {0} closed its control connection without a goodbye (process died?)".format(who)
                            self._deliver_abort(reason)
                            if self.rank == 0:
                                self._relay_abort(reason, skip=who)
                        if self.rank == 0:
                            for r, s in list(self._ctl_peers.items()):
                                if s is sock:
                                    del self._ctl_peers[r]
                        else:
                            return
                        continue
                    if header.get("kind") == "abort":
                        self._deliver_abort(header.get("reason", "unknown"))
                        if self.rank == 0:
                            who = next((r for r, s in self._ctl_peers.items() if s is sock), None)
                            self._relay_abort(header.get("reason", "unknown"), skip=who)
                    elif header.get("kind") == "bye":
                        if self.rank == 0:
                            for r, s in list(self._ctl_peers.items()):
                                if s is sock:
                                    del self._ctl_peers[r]
                        else:
                            return
        except Exception:
            return

    # ------------------------------------------------------------------ collectives
    def _exchange(self, kind, header, payload, combine):
        """Every rank contributes (header, payload); the hub calls ``combine(list of (header, payload) in rank order)``
        -> (header, payload) and sends the result to everybody."""
        if self.aborted is not None:
            raise HostGroupError("the job was aborted: " + self.aborted)
        self._seq += 1
        header = dict(header, kind=kind, seq=self._seq)
        if self.world == 1:
            return combine([(header, payload)])
        try:
            if self.rank == 0:
                parts = [(header, payload)]
                for r in range(1, self.world):
                    h, p = _recv(self._peers[r])
                    if h.get("kind") != kind or h.get("seq") != self._seq:
                        raise HostGroupError("collective mismatch: rank {0} is in {1}#{2}, rank 0 in {3}#{4}".format(
                            r, h.get("kind"), h.get("seq"), kind, self._seq))
                    parts.append((h, p))
                out_h, out_p = combine(parts)
                for r in range(1, self.world):
                    _send(self._peers[r], out_h, out_p)
                return out_h, out_p
            _send(self._hub, header, payload)
            return _recv(self._hub)
        except socket.timeout:
            raise HostGroupError("{0}: no answer within {1:.0f} s (rank {2})".format(kind, self.timeout, self.rank))
        except OSError as exc:
            raise HostGroupError("{0}: {1}".format(kind, exc))

    def barrier(self):
        self._exchange("barrier", {}, b"", lambda parts: ({}, b""))

    def bcast_bytes(self, data=None, src=0):
        def combine(parts):
            return {}, parts[src][1]
        return self._exchange("bcast", {}, bytes(data) if self.rank == src and data is not None else b"", combine)[1]

    def bcast_json(self, obj=None, src=0):
        """Any JSON-serialisable object from ``src`` to everybody."""
        raw = self.bcast_bytes(json.dumps(obj).encode() if self.rank == src else None, src=src)
        return json.loads(raw.decode())

    def bcast_array(self, array=None, src=0):
        """A float64 / int64 array from ``src`` to everybody (shape travels with it)."""
        if self.rank == src:
            a = np.ascontiguousarray(array)
            code = "i8" if a.dtype.kind in "iu" else "f8"
            a = a.astype(_DTYPES[code], copy=False)
            meta, raw = {"shape": list(a.shape), "dtype": code}, a.tobytes()
        else:
            meta, raw = {}, b""

        def combine(parts):
            return {k: parts[src][0][k] for k in ("shape", "dtype")}, parts[src][1]
        h, p = self._exchange("bcast_array", meta, raw, combine)
        return np.frombuffer(p, dtype=_DTYPES[h["dtype"]]).reshape(h["shape"]).copy()

    def allreduce(self, array, op="sum"):
        """Element-wise sum / max / min over the ranks of a float64 or int64 array (or scalar); same result on every
        rank, combined in rank order."""
        a = np.asarray(array, order="C")                    # (ascontiguousarray would turn a scalar into shape (1,))
        code = "i8" if a.dtype.kind in "iub" else "f8"
        a = a.astype(_DTYPES[code], copy=False)
        fn = {"sum": np.add, "max": np.maximum, "min": np.minimum}[op]

        def combine(parts):
            shapes = {tuple(h["shape"]) for h, _ in parts}
            if len(shapes) != 1 or len({h["dtype"] for h, _ in parts}) != 1:
                raise HostGroupError("allreduce: ranks passed different shapes / dtypes: {0}".format(sorted(shapes)))
            acc = np.frombuffer(parts[0][1], dtype=_DTYPES[code]).copy()
            for _, p in parts[1:]:
                acc = fn(acc, np.frombuffer(p, dtype=_DTYPES[code]))
            return {"shape": list(a.shape), "dtype": code}, acc.tobytes()
        h, p = self._exchange("allreduce_" + op, {"shape": list(a.shape), "dtype": code}, a.tobytes(), combine)
        out = np.frombuffer(p, dtype=_DTYPES[h["dtype"]]).reshape(h["shape"]).copy()
        return out if out.ndim else out[()]

    def allgather_array(self, array):
        """List (one entry per rank, rank order) of the ranks' float64 arrays; lengths may differ."""
        a = np.ascontiguousarray(array, dtype=np.float64)

        def combine(parts):
            return {"sizes": [len(p) for _, p in parts], "shapes": [h["shape"] for h, _ in parts]}, b"".join(p for _, p in parts)
        h, p = self._exchange("allgather", {"shape": list(a.shape)}, a.tobytes(), combine)
        out, off = [], 0
        for size, shape in zip(h["sizes"], h["shapes"]):
            out.append(np.frombuffer(p[off:off + size], dtype=np.float64).reshape(shape).copy())
            off += size
        return out

    def same_everywhere(self, array):
        """True when every rank passed bit-identical data (CRC-32 of the bytes, min == max over ranks)."""
        crc = zlib.crc32(np.ascontiguousarray(array).tobytes())
        both = self.allreduce(np.array([crc, -crc], dtype=np.int64), op="max")
        return int(both[0]) == -int(both[1])

    def close(self):
        self._closing = True
        for s in list(self._ctl_peers.values()) + [self._ctl_hub]:
            if s is not None:
                try:
                    _send(s, {"kind": "bye"})
                except (OSError, HostGroupError):
                    pass
                try:
                    s.close()
                except OSError:
                    pass
        self._ctl_peers, self._ctl_hub = {}, None
        for s in list(self._peers.values()) + [self._hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._hub, self._listener = {}, None, None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
