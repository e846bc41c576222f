"""Weight files for the drop-in models (reference FRVSR.py:88,193,212; TecoGAN.py:158,268,517,537).

The reference stores Keras 2.2.4 ``save_weights`` HDF5 files (``demo.h5``, checkpoints).  Their
layout: one group per top-level layer, only ``FNet`` (28 datasets) and
``SRNet`` (48 datasets) carry weights, dataset names ``<group>/conv2d_<n>/kernel:0`` etc., and
Keras matches them to layers BY ORDER, not by name.  This module keeps exactly that contract on
an ordered ``[(keras_name, float32 array)]`` list (names carry the group, ``FNet/conv2d_1/kernel:0``; inside
the file Keras stores ``weight_names`` WITHOUT it -- the variable name ``conv2d_1/kernel:0`` -- so the dataset
sits at ``FNet/conv2d_1/kernel:0``; the writer does the same, the reader accepts both) and supports two containers:

  * ``.h5`` / ``.hdf5``  the Keras layout, read/written with h5py when it is importable, else with the
                bundled ``h5lite`` (the subset of HDF5 that h5py's default format uses for such files;
                validated against a libhdf5-written sample, but no Keras-written file exists in this
                environment: PARITY UNPINNED for real checkpoints);
  * ``.npz``            the same names and arrays in a numpy archive plus ``__order__`` (file order).
"""
from __future__ import annotations

import os

import numpy as np

GROUPS = ("FNet", "SRNet")


def _group_of(name):
    return name.split("/")[0]


def read_tensors(path):
    """Ordered [(keras_name, array)] of every weight tensor in the file."""
    ext = os.path.splitext(path)[1].lower()
    if not os.path.exists(path):
        raise OSError(f"Unable to open weight file {path!r}: no such file")
    if ext == ".npz":
        with np.load(path, allow_pickle=False) as z:
            order = [str(n) for n in z["__order__"]] if "__order__" in z.files else [n for n in z.files]
            return [(n, np.asarray(z[n], np.float32)) for n in order if n != "__order__"]
    if ext in (".h5", ".hdf5"):
        try:
            import h5py as h5
        except ImportError:
            import h5lite as h5
        out = []
        with h5.File(path, "r") as f:
            g = f["model_weights"] if "model_weights" in f else f
            layer_names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]
            for ln in layer_names:
                lg = g[ln]
                for wn in lg.attrs.get("weight_names", []):
                    wn = wn.decode() if isinstance(wn, bytes) else wn
                    name = wn if wn.startswith(ln + "/") else ln + "/" + wn
                    out.append((name, np.asarray(lg[wn][()], np.float32)))
        return out
    raise ValueError(f"{path}: unsupported weight file extension {ext!r} (use .h5 or .npz)")


def _local(name):
    """Name of a weight inside its group, as Keras writes it: 'FNet/conv2d_1/kernel:0' -> 'conv2d_1/kernel:0'."""
    return name.split("/", 1)[1] if "/" in name else name


def write_tensors(path, tensors, weightless_layers=(), keep_layer_prefix=False, layer_order=None):
    """weightless_layers: names of extra top-level layers without weights (Keras lists every layer of the model in
    ``layer_names`` -- the inputs, Upscaling_0, Warp_1 ... -- each with an empty group).
    layer_order: every top-level layer name in the model's own order (overrides weightless_layers first / weighted after).
    keep_layer_prefix: the model's layers are plain layers, not sub-models (the discriminator): Keras names the weight
    ``conv2d_37/kernel:0`` and stores it INSIDE the group ``conv2d_37``, i.e. at ``conv2d_37/conv2d_37/kernel:0``."""
    tensors = [(n, np.ascontiguousarray(a, np.float32)) for n, a in tensors]
    _local = (lambda n: n) if keep_layer_prefix else globals()["_local"]
    ext = os.path.splitext(path)[1].lower()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if ext == ".npz":
        arrays = {n: a for n, a in tensors}
        arrays["__order__"] = np.array([n for n, _ in tensors])
        with open(path, "wb") as fh:
            np.savez(fh, **arrays)
        return
    if ext in (".h5", ".hdf5"):
        if layer_order is not None:           # the model's layers in Keras' own order, weighted and weightless interleaved
            groups = [str(g) for g in layer_order]
            missing = [_group_of(n) for n, _ in tensors if _group_of(n) not in groups]
            if missing:
                raise ValueError(f"layer_order lacks the weighted layers {sorted(set(missing))}")
        else:
            groups = [str(g) for g in weightless_layers]
            for n, _ in tensors:
                if _group_of(n) not in groups:
                    groups.append(_group_of(n))
        # a layer without weights: Keras 2.2.4 hands h5py an empty list, which h5py stores as an EMPTY float64 attribute
        # (save_attributes_to_hdf5_group -> np.asarray([])); both back ends write exactly that
        no_names = np.zeros((0,), np.float64)
        try:
            import h5py
        except ImportError:
            import h5lite
            root = {"attrs": {"layer_names": np.array([g.encode() for g in groups]), "backend": np.bytes_(b"tensorflow"),
                              "keras_version": np.bytes_(b"2.2.4")}, "children": {}}
            for g in groups:
                names = [_local(n) for n, _ in tensors if _group_of(n) == g]
                node = {"attrs": {"weight_names": np.array([n.encode() for n in names]) if names else no_names}, "children": {}}
                for n, a in tensors:
                    if _group_of(n) != g:
                        continue
                    cur = node                            # dataset path inside the group = the weight name, as Keras writes it
                    parts = _local(n).split("/")
                    for part in parts[:-1]:
                        cur = cur["children"].setdefault(part, {"attrs": {}, "children": {}})
                    cur["children"][parts[-1]] = a
                root["children"][g] = node
            h5lite.write_file(path, root)
            return
        with h5py.File(path, "w") as f:
            f.attrs["layer_names"] = [g.encode() for g in groups]
            f.attrs["backend"] = b"tensorflow"
            f.attrs["keras_version"] = b"2.2.4"
            for g in groups:
                lg = f.create_group(g)
                names = [_local(n) for n, _ in tensors if _group_of(n) == g]
                lg.attrs["weight_names"] = [n.encode() for n in names] if names else no_names
                for n, a in tensors:
                    if _group_of(n) == g:
                        lg.create_dataset(_local(n), data=a)
        return
    raise ValueError(f"{path}: unsupported weight file extension {ext!r} (use .h5 or .npz)")


def load_into(model, path):
    """Keras semantics: tensors of the groups this model owns, matched by position and shape."""
    tensors = read_tensors(path)
    want = model.named_weights() if hasattr(model, "named_weights") else []
    prefixes = []
    for n, _ in want:
        if _group_of(n) not in prefixes:
            prefixes.append(_group_of(n))
    have = [(n, a) for n, a in tensors if _group_of(n) in prefixes]
    if len(have) != len(want):
        raise ValueError(f"{path}: holds {len(have)} tensors for {prefixes}, the model {model.name} "
                         f"expects {len(want)}")
    ordered = []
    for p in prefixes:      # keep each group's own order, groups in model order
        ordered += [a for n, a in have if _group_of(n) == p]
    for (wn, wa), a in zip(want, ordered):
        if tuple(a.shape) != tuple(wa.shape):
            raise ValueError(f"{path}: tensor for {wn} has shape {tuple(a.shape)}, expected {tuple(wa.shape)}")
    model.set_weights(ordered)


def save_from(model, path):
    write_tensors(path, model.named_weights() if hasattr(model, "named_weights") else [])
