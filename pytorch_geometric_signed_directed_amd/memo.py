"""One memo type for every "same graph tensors as last call -> same operator" shortcut of the package.

The reference re-derives its operators (Laplacian, gcn_norm, sorted edge lists) on every forward unless
`cached=True`.  On the GPU that re-sort is the expensive part, so the layers keep their last few derived
operators and reuse them while the inputs are THE SAME TENSOR OBJECTS, unmodified.  Contract of a hit:

  * every key tensor is the same Python object (held by WEAK reference: the memo never keeps a user's
    `edge_index` / `edge_weight` alive, and an entry -- with the CSRs and value copies it holds -- is dropped as
    soon as one of its key tensors is garbage collected);
  * same in-place version counter (`tensor._version`), same storage address and shape (so `.set_()` /
    re-allocation are seen);
  * same non-tensor key (node count, q, normalisation ...).

Writes that bypass the version counter -- `edge_index.data[...] = ...`, `.data.copy_()`, raw-pointer writes of another
library or a custom kernel -- are invisible to the three checks above, and the reference, which re-derives everything
from the CURRENT contents on every uncached forward, would see them.  After such a write the default memo returns the
operator of the OLD contents.  Two ways out:

  * STRICT MODE (round 6), `PYGSD_MEMO_VERIFY=1` / `memo.set_verify(True)`: hits are verified by CONTENT.  A layer that is
    about to consult its memos fingerprints its graph tensors on the device (pygsd_fingerprint_u64: 64 bits over the bytes,
    ~0.04 ms for a 20 M-edge edge_index), compares them with the fingerprints it took of the SAME tensor objects the last
    time -- one device -> host read for all of them (`verified`) -- and, where one differs, forgets every memoised value
    derived from that tensor before anything is looked up.  The layers then behave like the reference after ANY write, and
    a hit still saves the rebuild (0.94 ms at the north star).  Measured cost (tools/memo_verify_probe.py): nothing visible
    on the north-star step (6.80 vs 6.78 ms), +0.17 ms on SIMPA's 2.75 ms step -- the host read takes the host's run-ahead
    away once per uncached forward, which is why it is a switch and not the default.  A cached=True layer that holds its
    operator never asks (the reference does not rebuild there either).
  * the opt-outs below, or `memo.clear_all()` after the write.

Switches:
    PYGSD_MEMO_VERIFY=1 / memo.set_verify(True)    strict mode: content-verified hits (above)
    PYGSD_NO_OPERATOR_MEMO=1                        (environment, whole process) no memo at all
    memo.set_enabled(False)                         (whole process, at run time)
    Layer(..., operator_memo=False)                 (one layer; MagNetConv / MSConv / DGCNConv / Conv_Base)

Lookups and insertions are serialised by a lock (DataLoader threads).  Streams: an entry remembers the HIP stream it was last
produced / used on; a hit from another stream makes that stream wait for it first (`_order_after`).
"""
import os
import threading
import weakref
from typing import Any, Hashable, Optional, Sequence

import torch

_enabled = os.environ.get("PYGSD_NO_OPERATOR_MEMO", "0") in ("", "0")
_verify = os.environ.get("PYGSD_MEMO_VERIFY", "0") == "1"
_epoch = 0          # bumped whenever a content change is detected: caches outside TensorMemo (Pattern._vcache) compare it
_registry = weakref.WeakSet()
_registry_lock = threading.Lock()
_scope = threading.local()


def verify() -> bool:
    return _verify


def set_verify(on: bool) -> bool:
    """Content verification of memo keys (module docstring) on / off; returns the previous setting."""
    global _verify
    prev, _verify = _verify, bool(on)
    return prev


def enabled() -> bool:
    return _enabled


def set_enabled(on: bool) -> None:
    global _enabled
    _enabled = bool(on)
    if not _enabled:
        clear_all()


def clear_all() -> None:
    """Drop every memoised operator of every layer / module-level cache in the process."""
    with _registry_lock:
        memos = list(_registry)
    for m in memos:
        m.clear()


class _Entry:
    __slots__ = ("refs", "stamps", "extra", "value", "stream_ptr", "stream")

    def __init__(self, refs, stamps, extra, value, stream_ptr=None, stream=None):
        self.refs, self.stamps, self.extra, self.value = refs, stamps, extra, value
        self.stream_ptr, self.stream = stream_ptr, stream      # the HIP stream the value was last produced / used on


def _raw_stream(tensors):
    """Raw handle of the current HIP stream of the key tensors' device (None for host tensors): two integer calls into torch."""
    for t in tensors:
        if t is not None and t.is_cuda:
            return torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None else torch._C._cuda_getDevice())
    return None


def _order_after(entry, tensors):
    """A memoised value is device data queued on SOME stream -- the one its miss ran on, or the one a later hit finished a lazily
    built member on (a pattern's transposed CSR, a value array).  A caller on another stream has synchronised with the tensors
    it passes in, not with that stream: before it uses the entry, its stream waits for everything queued so far on the
    entry's last stream, and becomes the entry's stream.  Same stream (every single-stream program): one integer compare."""
    if entry.stream_ptr is None:
        return
    now = _raw_stream(tensors)
    if now is None or now == entry.stream_ptr:
        return
    if torch.cuda.is_current_stream_capturing():
        return      # a captured region cannot wait on work outside it; hipgraph.capture_step joins its warm-up stream before capturing
    dev = next(t.device for t in tensors if t is not None and t.is_cuda)
    cur = torch.cuda.current_stream(dev)
    done = torch.cuda.Event()
    done.record(entry.stream)
    cur.wait_event(done)
    entry.stream_ptr, entry.stream = now, cur


def _stamp(t: Optional[torch.Tensor]):
    return None if t is None else (t._version, t.data_ptr(), tuple(t.shape))


class TensorMemo:
    """Small LRU keyed on (tensor identities, their versions / storage, a hashable extra)."""

    def __init__(self, capacity: int = 4, on: Optional[bool] = None, verify: bool = True):
        self.capacity = max(int(capacity), 1)
        self.on = on                       # None: follow the process-wide switch
        # verify: a hit is only a hit if the key tensors' CONTENTS are what they were (check_unchanged; module docstring).
        # False for memos whose value does not depend on the contents in a way that could make a result wrong (which build
        # path to try first) or whose owner opted into trusting identity (the sharded layers' input-exchange memo).
        self.verify = bool(verify)
        self._items = []
        self._lock = threading.RLock()
        with _registry_lock:
            _registry.add(self)

    def active(self) -> bool:
        return _enabled if self.on is None else (bool(self.on) and _enabled)

    # A memo is a cache, not state: copying or pickling the module that owns it (copy.deepcopy for best-model
    # snapshots / swa_utils.AveragedModel, torch.save(model), mp.spawn arguments) yields a fresh, EMPTY memo with the
    # same capacity and switch -- never the lock, the weak references or another module's operators.
    def __deepcopy__(self, _memo_dict):
        return TensorMemo(self.capacity, self.on, self.verify)

    def __copy__(self):
        return TensorMemo(self.capacity, self.on, self.verify)

    def __reduce__(self):
        return (TensorMemo, (self.capacity, self.on, self.verify))

    def _prune(self, _ref=None):
        # called from weakref callbacks, possibly re-entrantly on the thread that is inside get() (a cyclic-GC pass
        # triggered by an allocation there): filter IN PLACE so that a list object held by the caller stays the list
        with self._lock:
            self._items[:] = [e for e in self._items if all(r is None or r() is not None for r in e.refs)]

    def get(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable = None, trusted: bool = False) -> Any:
        """trusted: the key tensors are the package's OWN derived tensors (a normalised or flipped edge list that no caller
        ever holds): nothing outside can have written them, no content check."""
        if not self.active():
            return None
        hit = self._lookup(tensors, extra)
        if hit is not None and self.verify and _verify and not trusted and check_unchanged(*tensors):
            return None                    # the contents changed behind the version counter: every entry on them is gone
        return None if hit is None else hit[0]

    def _lookup(self, tensors, extra):
        with self._lock:
            stamps = tuple(_stamp(t) for t in tensors)
            for e in list(self._items):                      # a snapshot: _prune may shorten the list meanwhile
                if e.extra != extra or len(e.refs) != len(tensors):
                    continue
                if all((r is None and t is None) or (r is not None and t is not None and r() is t)
                       for r, t in zip(e.refs, tensors)) and e.stamps == stamps:
                    try:                                     # move to the MRU end, by identity (never by a stale index)
                        self._items.remove(e)
                    except ValueError:
                        pass
                    self._items.append(e)
                    _order_after(e, tensors)                 # (a caller on another stream waits for the entry's)
                    return (e.value,)
        return None

    def put(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable, value: Any) -> Any:
        if self.active():
            if self.verify and _verify:
                check_unchanged(*tensors)      # strict mode: take the fingerprints of what this value was derived from (queued)
            owner = weakref.ref(self)

            def gone(_ref, owner=owner):
                m = owner()
                if m is not None:
                    m._prune()

            refs = tuple(None if t is None else weakref.ref(t, gone) for t in tensors)
            raw = _raw_stream(tensors)
            stream = None if raw is None else torch.cuda.current_stream(next(t.device for t in tensors if t is not None and t.is_cuda))
            with self._lock:
                self._items.append(_Entry(refs, tuple(_stamp(t) for t in tensors), extra, value, raw, stream))
                if len(self._items) > self.capacity:
                    self._items.pop(0)
        return value

    def clear(self) -> None:
        with self._lock:
            self._items[:] = []

    def forget(self, tensor: torch.Tensor) -> None:
        """Drop every entry one of whose key tensors is `tensor` (its contents changed behind the version counter)."""
        with self._lock:
            self._items[:] = [e for e in self._items if not any(r is not None and r() is tensor for r in e.refs)]

    def __len__(self) -> int:
        with self._lock:
            return len(self._items)


# ---- content verification ---------------------------------------------------------------------------------------------------
_FINGERPRINTS = TensorMemo(64, verify=False)   # tensor -> the device fingerprint taken when it was last checked (weakly held, per stamp)


def content_epoch() -> int:
    """Bumped whenever check_unchanged finds a changed tensor: value caches outside this module (sparse.Pattern's permuted
    copies of edge values) are valid for one epoch."""
    return _epoch


def _forget_everywhere(tensor: torch.Tensor) -> None:
    global _epoch
    _epoch += 1
    with _registry_lock:
        memos = list(_registry)
    for m in memos:
        m.forget(tensor)


_owned = {}         # id -> weak reference: tensors this package derived itself (normalised / flipped edge lists and their weights)


def own(*tensors: Optional[torch.Tensor]) -> None:
    """Mark tensors as the package's OWN derived data -- a normalised or flipped edge list, normalised weights: objects no
    caller ever holds, so nothing outside can have written them.  Lookups keyed on them are never content-checked."""
    for t in tensors:
        if isinstance(t, torch.Tensor):
            key = id(t)
            _owned[key] = weakref.ref(t, lambda _r, key=key: _owned.pop(key, None))


def _is_owned(t: torch.Tensor) -> bool:
    ref = _owned.get(id(t))
    return ref is not None and ref() is t


def _fingerprint(t: torch.Tensor) -> Optional[torch.Tensor]:
    """int64[1] fingerprint of a DEVICE tensor (queued, no host read); None for tensors that are not checked (host tensors:
    the package computes on the device only).  The one place the tests replace to exercise the logic below without a GPU."""
    if not t.is_cuda:
        return None
    from . import _cabi
    return _cabi.fingerprint(t)


def check_unchanged(*tensors: Optional[torch.Tensor]) -> int:
    """Fingerprint the given device tensors and compare with the fingerprints of the same objects at their last check; a tensor
    whose CONTENTS changed while identity, version and storage did not is forgotten by every memo in the process.  One device ->
    host read for all tensors together (none on a first sighting).  Returns the number of tensors found changed."""
    if not (_verify and _enabled):
        return 0
    done = getattr(_scope, "done", None)                     # tensors a surrounding `verified` scope has checked already
    seen, pairs = set(), []
    for t in tensors:
        if t is None or not isinstance(t, torch.Tensor) or t.numel() == 0 or id(t) in seen or _is_owned(t):
            continue
        if done is not None:
            if id(t) in done:
                continue
            done.add(id(t))
        seen.add(id(t))
        now = _fingerprint(t)
        if now is None:
            continue
        before = _FINGERPRINTS.get((t,), "fingerprint")
        if before is None:
            _FINGERPRINTS.put((t,), "fingerprint", now)      # first sighting, or a new version / storage: nothing to compare
        else:
            pairs.append((t, before, now))
    if not pairs:
        return 0
    values = torch.cat([b for _, b, _ in pairs] + [n for _, _, n in pairs]).tolist()      # the one host read
    k, changed = len(pairs), 0
    for j, (t, _, now) in enumerate(pairs):
        if values[j] != values[k + j]:
            changed += 1
            _forget_everywhere(t)                            # (drops the stale fingerprint as well)
            _FINGERPRINTS.put((t,), "fingerprint", now)
    return changed


def trust(*tensors: Optional[torch.Tensor]) -> None:
    """Inside a `verified` scope: mark the package's own derived tensors (a flipped edge list computed from a checked one) as
    checked, so that lookups keyed on them do not fingerprint them.  Outside a scope: nothing."""
    done = getattr(_scope, "done", None)
    if done is not None:
        done.update(id(t) for t in tensors if t is not None)


class verified:
    """`with memo.verified(edge_index, edge_weight): ...` -- check these tensors once (one host read for all of them), then a
    scope in which they are not checked again: SIMPA's seven Conv_Base calls on one edge_index, the three memos of a magnetic
    layer.  Any OTHER tensor looked up inside the scope is still checked (once).  Scopes nest; a scope lasts for the `with`
    block only -- user code cannot run inside it, so nothing can change the tensors in between."""

    def __init__(self, *tensors):
        self._tensors = tensors

    def __enter__(self):
        self._outer = getattr(_scope, "done", None)
        _scope.done = set() if self._outer is None else self._outer      # nested: share the outer scope's set
        check_unchanged(*self._tensors)
        return self

    def __exit__(self, *exc):
        _scope.done = self._outer
        return False

