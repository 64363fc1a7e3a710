"""A tensor whose contents are produced on first use.

`isect_tiles` returns `isect_ids` (8 B per intersection: 158 MB per S-1M frame) because gsplat does, but on the
reference's path nothing ever reads them: the only consumer is `isect_offset_encode` (renderer.py:253), whose
result the tile-bucketed path already has.  `LazyTensor` wraps the (allocated, not yet written) output buffer
and a `fill` callable; any torch operation that could observe the contents -- arithmetic, indexing, `.cpu()`,
`.numpy()`, `print`, `data_ptr()`, ... -- first runs `fill` (one kernel that rebuilds the keys from the sorted
lists) on the current stream.  Pure metadata (shape, dtype, device, numel, ...) does not.  After the fill the
object behaves like any other tensor.  Code that hands the raw pointer to a C extension WITHOUT going through a
torch call must call `.materialize()` itself; within this package only the operators do that."""
from __future__ import annotations

import torch

_METADATA_ONLY = {
    "numel", "size", "dim", "ndimension", "nelement", "element_size", "is_contiguous", "stride", "storage_offset",
    "is_floating_point", "is_complex", "is_signed", "get_device", "type", "is_pinned", "is_shared",
}


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
    def __new__(cls, buffer: torch.Tensor, fill):
        t = torch.Tensor._make_subclass(cls, buffer, False)
        t._sc_fill = fill
        return t

    def materialize(self) -> torch.Tensor:
        """Runs the pending fill (if any) and returns self."""
        fill = self.__dict__.get("_sc_fill")
        if fill is not None:
            self.__dict__["_sc_fill"] = None
            with torch._C.DisableTorchFunctionSubclass():
                fill(self)
        return self

    @property
    def is_materialized(self) -> bool:
        return self.__dict__.get("_sc_fill") is None

    # copies and serialisation hand out ORDINARY tensors with the contents in place: the pending fill is a closure over
    # this frame's device buffers and must not travel (copy.copy goes through torch's own path, which reads -- and so
    # fills -- the tensor; torch's default __deepcopy__ refuses non-wrapper subclasses)
    def __deepcopy__(self, memo):
        out = self.materialize().as_subclass(torch.Tensor).clone()
        memo[id(self)] = out
        return out

    def __reduce_ex__(self, proto):
        return self.materialize().as_subclass(torch.Tensor).__reduce_ex__(proto)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if not _is_metadata(func):
            for a in list(args) + list(kwargs.values()):
                if isinstance(a, LazyTensor):
                    a.materialize()
                elif isinstance(a, (list, tuple)):
                    for b in a:
                        if isinstance(b, LazyTensor):
                            b.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        # results are ordinary tensors (views of the filled buffer included)
        if isinstance(out, LazyTensor) and out is not (args[0] if args else None):
            out = out.as_subclass(torch.Tensor)
        return out
