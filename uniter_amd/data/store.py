"""Key -> bytes record stores behind the datasets.

The reference keeps its text and image records in LMDB environments (data/data.py:68-72,140-151).  Two stores are offered:

  * `LmdbStore` — the reference's databases as they are; needs the `lmdb` package (absent from this image: opening one
    raises ImportError with that explanation, it never silently falls back).
  * `PackStore` — a directory with `index.json` + `data.bin`: the records back to back in one file that is memory-mapped
    once; `get()` returns a zero-copy view.  Same contents, no B-tree and no page cache double-buffering; this is what the
    tests, the tools and a node with the whole corpus in RAM use.  `PackWriter` builds one, `convert_store` copies any store
    (e.g. an LMDB) into it.

`FeaturePack` goes one step further for the image side (the survey's point: at ~20 k examples/s per GPU the CPU-side
decode of 36 x 2048 features, not PCIe, is the bottleneck): all images' arrays concatenated along the box axis in flat
fp16 files, memory-mapped; an image is a row range — no inflate, no unpickle, no copy until the collate writes the padded
batch (into pinned memory, which PrefetchLoader then sends and casts to bf16 on the copy stream)."""
import json
import mmap
import os

import numpy as np


class LmdbStore(object):
    """Read-only (or, with readonly=False, write) view of an LMDB environment, keys as str."""

    def __init__(self, path, readonly=True, readahead=True):
        try:
            import lmdb
        except ImportError as e:
            raise ImportError("opening %r needs the `lmdb` package; convert the database to a PackStore "
                              "(uniter_amd.data.store.convert_store) or install lmdb" % (path,)) from e
        self.path = path
        self.readonly = readonly
        if readonly:
            self.env = lmdb.open(path, readonly=True, create=False, readahead=readahead)
            self.txn = self.env.begin(buffers=True)
        else:
            self.env = lmdb.open(path, readonly=False, create=True, map_size=4 * 1024 ** 4)
            self.txn = self.env.begin(write=True)
        self._writes = 0

    def get(self, key):
        return self.txn.get(key.encode('utf-8'))

    def put(self, key, blob):
        if self.readonly:
            raise ValueError("read-only store")
        self.txn.put(key.encode('utf-8'), bytes(blob))
        self._writes += 1
        if self._writes % 1000 == 0:             # (the reference commits every 1000 records too)
            self.txn.commit()
            self.txn = self.env.begin(write=True)

    def keys(self):
        with self.env.begin() as txn:
            return [bytes(k).decode('utf-8') for k, _ in txn.cursor()]

    def close(self):
        if self.env is not None:
            if not self.readonly:
                self.txn.commit()
            self.env.close()
            self.env = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackStore(object):
    """Read side of a pack directory: {index.json: {"version": 1, "records": {key: [offset, nbytes]}}, data.bin}."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, 'index.json')) as f:
            index = json.load(f)
        if index.get('version') != 1:
            raise ValueError("%s: unknown pack version %r" % (path, index.get('version')))
        self._records = index['records']
        self._file = open(os.path.join(path, 'data.bin'), 'rb')
        size = os.fstat(self._file.fileno()).st_size
        self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ) if size else None
        self._view = memoryview(self._map) if self._map is not None else memoryview(b'')

    def get(self, key):
        entry = self._records.get(key)
        if entry is None:
            return None
        return self._view[entry[0]:entry[0] + entry[1]]

    def keys(self):
        return list(self._records.keys())

    def __len__(self):
        return len(self._records)

    def close(self):
        # slices handed out by get() may still be alive: memoryview.release() / mmap.close() then raise BufferError.  Drop our
        # references in that case (the mapping is unmapped when the last slice dies) instead of leaving the object half closed.
        if self._view is not None:
            try:
                self._view.release()
            except BufferError:
                pass
            self._view = None
        if self._map is not None:
            try:
                self._map.close()
            except BufferError:
                pass
            self._map = None
        if self._file is not None:
            self._file.close()
            self._file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackWriter(object):
    """with PackWriter(path) as w: w.put(key, blob) ...   (records are 64-byte aligned in data.bin)"""

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.path = path
        self._records = {}
        self._file = open(os.path.join(path, 'data.bin'), 'wb')
        self._pos = 0

    def put(self, key, blob):
        blob = bytes(blob)
        pad = (-self._pos) % 64
        if pad:
            self._file.write(b'\0' * pad)
            self._pos += pad
        self._records[key] = [self._pos, len(blob)]
        self._file.write(blob)
        self._pos += len(blob)

    def close(self):
        if self._file is None:
            return
        self._file.close()
        self._file = None
        with open(os.path.join(self.path, 'index.json'), 'w') as f:
            json.dump({'version': 1, 'records': self._records}, f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_store(path, readonly=True, readahead=True):
    """A pack directory if `path` holds one, else an LMDB environment."""
    if os.path.isfile(os.path.join(path, 'index.json')) and os.path.isfile(os.path.join(path, 'data.bin')):
        if not readonly:
            raise ValueError("pack stores are written with PackWriter")
        return PackStore(path)
    return LmdbStore(path, readonly=readonly, readahead=readahead)


def convert_store(src, dst_path, keys=None):
    """Copies every record (or `keys`) of an open store into a new pack directory; returns the number of records."""
    n = 0
    with PackWriter(dst_path) as w:
        for key in (keys if keys is not None else src.keys()):
            blob = src.get(key)
            if blob is not None:
                w.put(key, blob)
                n += 1
    return n


class FeaturePack(object):
    """Decode-free image features: `<path>/feature_pack.json` + one flat little-endian file per field.

    json: {"version": 1, "fields": {name: {"file": ..., "dtype": "<f2", "width": W}}, "images": {fname: [row0, rows]}}
    An image's arrays are rows [row0, row0 + rows) of every field file viewed as [total_rows, W]."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, 'feature_pack.json')) as f:
            meta = json.load(f)
        if meta.get('version') != 1:
            raise ValueError("%s: unknown feature pack version" % path)
        self.images = meta['images']
        self.fields = {}
        for name, spec in meta['fields'].items():
            width = int(spec['width'])
            file = os.path.join(path, spec['file'])
            rows = os.path.getsize(file) // (np.dtype(spec['dtype']).itemsize * max(width, 1))
            shape = (rows, width) if width else (rows,)
            self.fields[name] = np.memmap(file, dtype=np.dtype(spec['dtype']), mode='r', shape=shape) if rows else \
                np.zeros(shape, dtype=np.dtype(spec['dtype']))

    def __contains__(self, fname):
        return fname in self.images

    def rows(self, fname):
        return self.images[fname][1]

    def get(self, fname, field, limit=None):
        """Zero-copy [rows(, W)] view of one field of one image (first `limit` rows)."""
        row0, rows = self.images[fname]
        if limit is not None:
            rows = min(rows, int(limit))
        return self.fields[field][row0:row0 + rows]

    @staticmethod
    def build(path, images, dtype='<f2'):
        """images: iterable of (fname, {field: ndarray [rows(, W)]}); every image has the same fields and widths."""
        os.makedirs(path, exist_ok=True)
        files, specs, index = {}, {}, {}
        row = 0
        try:
            for fname, arrays in images:
                rows = None
                for name, arr in arrays.items():
                    arr = np.asarray(arr)
                    width = int(arr.shape[1]) if arr.ndim == 2 else 0
                    if name not in files:
                        files[name] = open(os.path.join(path, name + '.bin'), 'wb')
                        specs[name] = {'file': name + '.bin', 'dtype': dtype, 'width': width}
                    if specs[name]['width'] != width:
                        raise ValueError("field %r changes width at image %r" % (name, fname))
                    if rows is None:
                        rows = int(arr.shape[0])
                    elif rows != int(arr.shape[0]):
                        raise ValueError("fields of image %r disagree on the number of boxes" % fname)
                    files[name].write(np.ascontiguousarray(arr, dtype=np.dtype(dtype)).tobytes())
                index[fname] = [row, rows or 0]
                row += rows or 0
        finally:
            for f in files.values():
                f.close()
        with open(os.path.join(path, 'feature_pack.json'), 'w') as f:
            json.dump({'version': 1, 'fields': specs, 'images': index}, f)
        return FeaturePack(path)
