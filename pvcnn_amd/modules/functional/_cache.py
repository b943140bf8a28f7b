"""Per-tensor memo of coordinate-derived products.

Every PVConv of a network sees the SAME coords tensor object (the reference's models pass `coords` through
unchanged: models/s3dis/pvcnn.py:38-42, modules/pvconv.py:39), and everything the voxel branch derives from
coordinates -- the centred statistics, the float grid coordinates / integer voxel ids per resolution, the
counting-sort plans of the two scatters, the trilinear corner indices / weights -- depends on (coords, R) only,
not on the layer.  PVCNN has three PVConvs at R = 16 (PVCNN++ two or three per stage): they share one set.

Entries are keyed by the identity of a live tensor object and its in-place version counter; they die with the
tensor (weak reference), so nothing outlives the forward/backward pass that created it and an address reused
by a later tensor can never hit a stale entry.
"""
import threading
import weakref

__all__ = ['memo', 'forget', 'clear', 'enabled', 'tag_amax', 'amax_of']

# re-entrant: a weak-reference callback (_drop) can fire from the garbage collector at any allocation, including inside
# memo()'s own critical section on the same thread -- a plain Lock deadlocks there (seen in the GPU test-suite)
_lock = threading.RLock()
_entries = {}          # id(tensor) -> (weakref, (version, data_ptr, shape, stride), dict)
enabled = True


def _drop(key, ref):
    with _lock:
        cur = _entries.get(key)
        if cur is not None and cur[0] is ref:
            del _entries[key]


def _stamp(tensor):
    """What must be unchanged for a memo entry of `tensor` to be valid: the in-place version counter AND where / how the tensor
    views its storage (`t.data = other` or `set_()` re-seat a tensor without touching the counter)."""
    return (tensor._version, tensor.data_ptr(), tuple(tensor.shape), tuple(tensor.stride()))


def memo(tensor, key, make):
    """make() computed once per (tensor object, its version / storage view, key); the result is shared afterwards."""
    if not enabled:
        return make()
    tid = id(tensor)
    with _lock:
        ent = _entries.get(tid)
        if ent is not None and (ent[0]() is not tensor or ent[1] != _stamp(tensor)):
            ent = None
        if ent is None:
            ref = weakref.ref(tensor, lambda r, k=tid: _drop(k, r))
            ent = (ref, _stamp(tensor), {})
            _entries[tid] = ent
        store = ent[2]
        if key in store:
            return store[key]
    value = make()
    with _lock:
        store.setdefault(key, value)
        return store[key]


def forget(tensor, key):
    """Drop one memoised product of `tensor` (its owner found it stale)."""
    with _lock:
        ent = _entries.get(id(tensor))
        if ent is not None and ent[0]() is tensor:
            ent[2].pop(key, None)


def clear():
    with _lock:
        _entries.clear()


# ---- the f16x2 scale table ("amax buffer", include/pvcnn_hip.h) of a tensor, handed from the kernel that wrote the tensor to the
# kernels that consume it ----------------------------------------------------------------------------------------------------------
# The f16x2 products take their power-of-two operand scales from the bit patterns of max |x| per position segment (csrc/split16.h).
# A producer that touches every element anyway (the BatchNorm apply passes, forward and backward) emits the buffer for free; the
# consumer -- the next module / the next autograd node, which receives the very same tensor object -- finds it here instead of
# re-reading the tensor.  Keyed like every memo entry by (object identity, in-place version) plus the segment length: a tensor that
# was modified, accumulated into or replaced simply misses, and so does a consumer that wants another segmentation.
def tag_amax(tensor, seg, amax):
    if enabled and amax is not None:
        memo(tensor, ('amax', int(seg)), lambda: amax)
    return tensor


def amax_of(tensor, seg):
    """The amax buffer its producer left on `tensor` for segments of `seg` positions, or None."""
    if enabled:
        tid, key = id(tensor), ('amax', int(seg))
        with _lock:
            ent = _entries.get(tid)
            if ent is not None and ent[0]() is tensor and ent[1] == _stamp(tensor) and key in ent[2]:
                return ent[2][key]
    return None
