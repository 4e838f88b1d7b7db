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

What it cannot see: a write that bypasses the version counter -- `edge_index.data[...] = ...`, `.data.copy_()`,
a write through an external alias created with `.detach()` under `torch.no_grad()` does bump the counter, but
raw-pointer writes (another library, a custom kernel) do not.  After such a write the memo would return the
operator of the OLD contents where the reference rebuilds.  Opt out with any of

    PYGSD_NO_OPERATOR_MEMO=1            (environment, whole process)
    memo.set_enabled(False)             (whole process, at run time)
    Layer(..., operator_memo=False)     (one layer; MagNetConv / MSConv / DGCNConv / Conv_Base)

or call `memo.clear_all()` after the write.  Lookups and insertions are serialised by a lock (DataLoader threads,
multi-stream callers).
"""
import os
import threading
import weakref
from typing import Any, Hashable, Optional, Sequence

import torch

_enabled = os.environ.get("PYGSD_NO_OPERATOR_MEMO", "0") in ("", "0")
_registry = weakref.WeakSet()
_registry_lock = threading.Lock()


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
    __slots__ = ("refs", "stamps", "extra", "value")

    def __init__(self, refs, stamps, extra, value):
        self.refs, self.stamps, self.extra, self.value = refs, stamps, extra, value


def _stamp(t: Optional[torch.Tensor]):
    return None if t is None else (t._version, t.data_ptr(), tuple(t.shape))


class TensorMemo:
    """Small LRU keyed on (tensor identities, their versions / storage, a hashable extra)."""

    def __init__(self, capacity: int = 4, on: Optional[bool] = None):
        self.capacity = max(int(capacity), 1)
        self.on = on                       # None: follow the process-wide switch
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
        return TensorMemo(self.capacity, self.on)

    def __copy__(self):
        return TensorMemo(self.capacity, self.on)

    def __reduce__(self):
        return (TensorMemo, (self.capacity, self.on))

    def _prune(self, _ref=None):
        # called from weakref callbacks, possibly re-entrantly on the thread that is inside get() (a cyclic-GC pass
        # triggered by an allocation there): filter IN PLACE so that a list object held by the caller stays the list
        with self._lock:
            self._items[:] = [e for e in self._items if all(r is None or r() is not None for r in e.refs)]

    def get(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable = None) -> Any:
        if not self.active():
            return None
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
                    return e.value
        return None

    def put(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable, value: Any) -> Any:
        if self.active():
            owner = weakref.ref(self)

            def gone(_ref, owner=owner):
                m = owner()
                if m is not None:
                    m._prune()

            refs = tuple(None if t is None else weakref.ref(t, gone) for t in tensors)
            with self._lock:
                self._items.append(_Entry(refs, tuple(_stamp(t) for t in tensors), extra, value))
                if len(self._items) > self.capacity:
                    self._items.pop(0)
        return value

    def clear(self) -> None:
        with self._lock:
            self._items[:] = []

    def __len__(self) -> int:
        with self._lock:
            return len(self._items)
