"""A tensor whose contents are produced on first use.

`isect_tiles` returns `isect_ids` (8 B per intersection: 158 MB per S-1M frame) because gsplat does, but on the
reference's path nothing ever reads them: the only consumer is `isect_offset_encode` (renderer.py:253), whose
result the tile-bucketed path already has.  `LazyTensor` wraps the (allocated, not yet written) output buffer
and a `fill` callable; any torch operation that could observe the contents -- arithmetic, indexing, `.cpu()`,
`.numpy()`, `print`, `data_ptr()`, ... -- first runs `fill` (one kernel that rebuilds the keys from the sorted
lists) on the current stream.  Pure metadata (shape, dtype, device, numel, ...) does not.  After the fill the
object behaves like any other tensor.  Code that hands the raw pointer to a C extension WITHOUT going through a
torch call must call `.materialize()` itself; within this package only the operators do that.

Round 3: a LazyTensor may also have a pending RESOLVE step that comes before everything else, the SHAPE included.
`isect_tiles` enqueues its scatter + sort with predicted buffer sizes and would then have to wait for the frame's
intersection count only to hand out `flatten_ids` / `isect_ids` of the right length -- a host wait in the middle of
every frame, although the one consumer of those tensors on the reference's path (`rasterize_to_pixels`,
renderer.py:267) is three torch calls away.  With a resolver the wait moves to the first OBSERVATION of the tensor
(any torch call or attribute that depends on the shape or the contents: `shape`, `numel()`, `len()`, indexing,
arithmetic, `print`, `data_ptr()`, ...): the resolver re-points the object at the buffer of the true length
(`Tensor.set_`, in place: it is the same Python object the caller holds).  What does not depend on the shape -- dtype,
device, ndim (1), requires_grad -- is answered without resolving.  Code that unwraps the tensor in C++ without a torch
call (a pybind11 extension) sees the placeholder: call `.materialize()` / `.resolve_shape()` first; the operators of
this package do."""
from __future__ import annotations

import torch

_METADATA_ONLY = {
    "numel", "size", "dim", "ndimension", "nelement", "element_size", "is_contiguous", "stride", "storage_offset",
    "is_floating_point", "is_complex", "is_signed", "get_device", "type", "is_pinned", "is_shared", "__len__",
}


# what can be answered about a 1-D tensor whose LENGTH is still unknown
_SHAPE_FREE = {"dim", "ndimension", "element_size", "is_contiguous", "stride", "storage_offset", "is_floating_point",
               "is_complex", "is_signed", "get_device", "type", "is_pinned", "is_shared"}
_SHAPE_FREE_GETTERS = ("dtype", "device", "is_cuda", "requires_grad", "ndim", "layout", "is_sparse", "is_quantized",
                       "is_meta", "names", "_version", "grad_fn", "is_leaf", "grad", "is_cpu")


def _is_shape_free(func) -> bool:
    name = getattr(func, "__name__", "")
    if name in _SHAPE_FREE:
        return True
    if name == "__get__":
        return getattr(getattr(func, "__self__", None), "__name__", "") in _SHAPE_FREE_GETTERS
    return False


def _is_metadata(func) -> bool:
    name = getattr(func, "__name__", "")
    if name in _METADATA_ONLY:
        return True
    # property getters arrive as <method-wrapper '__get__' of getset_descriptor 'shape' ...>
    if name == "__get__":
        owner = getattr(func, "__self__", None)
        return getattr(owner, "__name__", "") in ("shape", "dtype", "device", "is_cuda", "requires_grad", "ndim",
                                                    "layout", "is_sparse", "is_quantized", "is_meta", "names",
                                                    "_version", "grad_fn", "is_leaf", "grad", "is_cpu")
    return False


class LazyTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, buffer: torch.Tensor, fill, resolve=None):
        """buffer: the storage handed out (its shape is a placeholder while `resolve` is pending); fill(self): writes
        the contents (or None); resolve(self): settles the shape, re-pointing self with Tensor.set_ (or None)."""
        t = torch.Tensor._make_subclass(cls, buffer, False)
        t._sc_fill = fill
        t._sc_resolve = resolve
        return t

    def resolve_shape(self) -> torch.Tensor:
        """Runs the pending resolver (if any): afterwards shape and storage are final.  Returns self."""
        resolve = self.__dict__.get("_sc_resolve")
        if resolve is not None and not self.__dict__.get("_sc_busy"):
            # the resolver is dropped only AFTER it has run: if it raises (more than 2^31 - 1 intersections, an allocation or
            # HIP error in the exact-size relaunch) it stays, and EVERY later observation raises again instead of handing out
            # the zero-length placeholder as if it were the list
            self.__dict__["_sc_busy"] = True
            try:
                with torch._C.DisableTorchFunctionSubclass():
                    resolve(self)
            finally:
                self.__dict__["_sc_busy"] = False
            self.__dict__["_sc_resolve"] = None
        return self

    def materialize(self) -> torch.Tensor:
        """Settles the shape, runs the pending fill (if any) and returns self."""
        self.resolve_shape()
        fill = self.__dict__.get("_sc_fill")
        if fill is not None and not self.__dict__.get("_sc_busy"):
            self.__dict__["_sc_busy"] = True
            try:
                with torch._C.DisableTorchFunctionSubclass():
                    fill(self)
            finally:
                self.__dict__["_sc_busy"] = False
            if self.__dict__.get("_sc_fill") is fill:       # (a fill that failed stays pending, like the resolver)
                self.__dict__["_sc_fill"] = None
        return self

    def _observe(self, contents: bool):
        if contents:
            self.materialize()
            # somebody has had their hands on the contents (a read, or an in-place write: they arrive alike): whatever was
            # cached ABOUT the contents on this object (isect_tiles files isect_offset_encode's result here) is dropped
            self.__dict__["_sc_touched"] = True
        else:
            self.resolve_shape()

    @property
    def untouched(self) -> bool:
        """True while no torch operation has seen the contents (reads and in-place writes alike)."""
        return not self.__dict__.get("_sc_touched", False)

    def plain(self) -> torch.Tensor:
        """The materialized contents as an ordinary tensor (same storage)."""
        self.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    @property
    def is_resolved(self) -> bool:
        return self.__dict__.get("_sc_resolve") is None

    @property
    def is_materialized(self) -> bool:
        return self.__dict__.get("_sc_resolve") is None and self.__dict__.get("_sc_fill") is None

    # copies and serialisation hand out ORDINARY tensors with the contents in place: the pending fill is a closure over
    # this frame's device buffers and must not travel (copy.copy goes through torch's own path, which reads -- and so
    # fills -- the tensor; torch's default __deepcopy__ refuses non-wrapper subclasses)
    def __deepcopy__(self, memo):
        out = self.materialize().as_subclass(torch.Tensor).clone()
        memo[id(self)] = out
        return out

    # DLPack: torch.from_dlpack(t) / cupy.from_dlpack(t) / ... ask the OBJECT for its capsule, so the contents can be settled
    # first.  (The legacy torch.utils.dlpack.to_dlpack(t) unwraps the tensor in C++ without any hook: it would export the
    # placeholder -- call t.materialize() first, or switch the deferral off: SC_DEFER_ISECT=0 SC_LAZY_IDS=0, INTEGRATION.md.)
    def __dlpack__(self, *args, **kwargs):
        self.materialize()
        self.__dict__["_sc_touched"] = True
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__dlpack__(*args, **kwargs)

    def __reduce_ex__(self, proto):
        return self.materialize().as_subclass(torch.Tensor).__reduce_ex__(proto)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if not _is_shape_free(func):
            contents = not _is_metadata(func)
            for a in list(args) + list(kwargs.values()):
                if isinstance(a, LazyTensor):
                    a._observe(contents)
                elif isinstance(a, (list, tuple)):
                    for b in a:
                        if isinstance(b, LazyTensor):
                            b._observe(contents)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        # results are ordinary tensors (views of the filled buffer included)
        if isinstance(out, LazyTensor) and out is not (args[0] if args else None):
            out = out.as_subclass(torch.Tensor)
        return out
