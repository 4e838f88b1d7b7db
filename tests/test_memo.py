"""CPU: pytorch_geometric_signed_directed_amd/memo.py -- the one memo type behind every "same graph tensors as last
call -> same operator" shortcut: hits only for the same tensor objects at the same version / storage / shape and the
same extra key, weak keys (an entry dies with its tensor), LRU capacity, per-instance and process-wide opt-outs,
and concurrent use."""
import gc
import threading

import torch

from pytorch_geometric_signed_directed_amd import memo
from pytorch_geometric_signed_directed_amd.memo import TensorMemo


def test_hit_needs_identity_version_storage_and_extra():
    m = TensorMemo(4)
    a, w = torch.arange(6).view(2, 3), torch.rand(3)
    m.put((a, w), ("n", 5), "op")
    assert m.get((a, w), ("n", 5)) == "op"
    assert m.get((a, w), ("n", 6)) is None                    # other extra key
    assert m.get((a.clone(), w), ("n", 5)) is None            # equal content, another object
    assert m.get((a, None), ("n", 5)) is None and m.get((a,), ("n", 5)) is None
    w.mul_(2)                                                 # in-place edit bumps the version
    assert m.get((a, w), ("n", 5)) is None
    m.put((a, w), ("n", 5), "op2")
    a.set_(torch.arange(6, 12).view(2, 3).untyped_storage(), 0, (2, 3), (3, 1))    # re-pointed storage, same object
    assert m.get((a, w), ("n", 5)) is None                    # ... seen through the storage address
    b = torch.zeros(4)
    m.put((b,), None, "b")
    b.data.add_(1)                                            # the documented blind spot: .data bypasses the version
    assert m.get((b,), None) == "b"


def test_entries_die_with_their_key_tensors_and_lru_evicts():
    m = TensorMemo(2)
    keep = [torch.zeros(1) for _ in range(3)]
    for k, t in enumerate(keep):
        m.put((t,), 0, k)
    assert len(m) == 2 and m.get((keep[0],), 0) is None and m.get((keep[2],), 0) == 2
    assert m.get((keep[1],), 0) == 1                          # touch -> most recent
    m.put((keep[0],), 0, 0)
    assert m.get((keep[2],), 0) is None and m.get((keep[1],), 0) == 1
    del keep[1]
    gc.collect()
    assert len(m) == 1                                        # the entry of the collected tensor is gone


def test_opt_outs():
    off = TensorMemo(2, on=False)
    t = torch.zeros(2)
    assert off.put((t,), 0, "v") == "v" and off.get((t,), 0) is None and len(off) == 0
    on = TensorMemo(2)
    on.put((t,), 0, "v")
    try:
        memo.set_enabled(False)                               # also clears every live memo
        assert not memo.enabled() and len(on) == 0 and on.get((t,), 0) is None
        on.put((t,), 0, "v")
        assert len(on) == 0
    finally:
        memo.set_enabled(True)
    on.put((t,), 0, "v")
    memo.clear_all()
    assert len(on) == 0


def test_concurrent_put_get_keeps_the_structure_consistent():
    m = TensorMemo(8)
    tensors = [torch.zeros(1) for _ in range(32)]
    errors = []

    def worker(seed):
        try:
            for k in range(2000):
                t = tensors[(seed * 7 + k) % len(tensors)]
                if m.get((t,), 0) is None:
                    m.put((t,), 0, id(t))
                else:
                    assert m.get((t,), 0) in (None, id(t))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors and len(m) <= 8
