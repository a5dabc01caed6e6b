"""Device-resident history buffers.

The reference appends Python lists per step (Agent.py:509-521, Neurons.py:681-687).
Here a history is a list of preallocated device chunks `[T_chunk, rows..., B]`
the kernels write in place; `stack()` returns the filled part as one tensor and
host copies are made only on demand (`get_history_arrays`, `history[...]`)."""
import torch


class DeviceHistory:
    def __init__(self, row_shape, dtype, device, chunk_bytes=1 << 30):
        self.row_shape = tuple(int(x) for x in row_shape)
        self.dtype = dtype
        self.device = device
        row_elems = 1
        for x in self.row_shape:
            row_elems *= x
        self.row_bytes = max(1, row_elems * torch.empty((), dtype=dtype).element_size())
        self.chunk_rows = int(max(1, min(4096, chunk_bytes // self.row_bytes)))
        self.chunks = []   # tensors [rows, *row_shape]
        self.filled = []   # rows used per chunk
        self.caps = []     # rows of each chunk (chunk.shape[0] builds a torch.Size: 0.3 us on a path that counts them)
        self.version = 0

    def __len__(self):
        return sum(self.filled)

    def reserve(self, T):
        """A writable view `[T, *row_shape]` of fresh rows (contiguous).  A request
        that does not fit the current chunk's free tail opens a new chunk of
        exactly max(T, chunk_rows) rows."""
        chunk, s = self.reserve_at(T)
        return chunk[s:s + int(T)]

    def reserve_at(self, T):
        """reserve() without making the view: (chunk tensor, first row).  Callers on a latency-critical path take
        the address as `chunk.data_ptr() + first_row * row_bytes` and build the view after their launch."""
        T = int(T)
        self.version += 1
        filled = self.filled
        if filled:
            s = filled[-1]
            if self.caps[-1] - s >= T:
                filled[-1] = s + T
                return self.chunks[-1], s
        rows = max(T, self.chunk_rows) if T == 1 else T
        self._new_chunk(rows)
        filled[-1] = T
        return self.chunks[-1], 0

    def _new_chunk(self, rows):
        self.chunks.append(torch.empty((rows, *self.row_shape), dtype=self.dtype, device=self.device))
        self.filled.append(0)
        self.caps.append(int(rows))

    def unreserve(self, T):
        """Give back the T rows of the latest reserve() (nothing was written to them)."""
        self.filled[-1] -= int(T)
        assert self.filled[-1] >= 0
        self.version += 1

    def preallocate(self, T):
        """Make sure the next `reserve(T)` finds T free rows (allocation happens here, not
        in the stepping loop)."""
        T = int(T)
        if self.chunks and self.caps[-1] - self.filled[-1] >= T:
            return
        self._new_chunk(T)

    def free_rows(self):
        """Rows left in the current chunk (0: the next reservation opens a new chunk)."""
        return self.caps[-1] - self.filled[-1] if self.chunks else 0

    def open_rows(self, T, chunk_rows=None):
        """A writable view of UP TO T free rows that are NOT yet counted as history (a step plan writes them one by
        one); `commit(n)` then publishes the first n of them.  A non-empty free tail of the current chunk is handed
        out as it is, however short (the caller re-opens when it is used up): a plan that is closed and rebuilt —
        the automatic stepper does that whenever a non-plain call interrupts the loop — continues in the rows its
        predecessor left instead of abandoning them behind a new chunk.  Only a full chunk is followed by a new one
        of max(T, chunk_rows) rows."""
        T = int(T)
        free = self.free_rows()
        if free <= 0:
            # `chunk_rows`: the size of a NEW chunk when T was cut down to another history's short tail (a plan opens
            # the same number of rows in every history): the new chunk is full-sized, only the view is T rows — a
            # 3-row tail elsewhere must not leave a 3-row chunk here (stack() concatenates every chunk on every read)
            rows = max(T, int(chunk_rows or T))
            self._new_chunk(rows)
            free = rows
        s = self.filled[-1]
        return self.chunks[-1][s:s + min(T, free)]

    def commit(self, n):
        if n:
            self.filled[-1] += int(n)
            assert self.filled[-1] <= self.caps[-1]
            self.version += 1

    def stack(self):
        """All filled rows as one tensor `[T_total, *row_shape]`."""
        parts = [c[:f] for c, f in zip(self.chunks, self.filled) if f]
        if not parts:
            return torch.empty((0, *self.row_shape), dtype=self.dtype, device=self.device)
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

    def row(self, k):
        """Filled row k (0 = oldest), as a view."""
        for c, f in zip(self.chunks, self.filled):
            if k < f:
                return c[k]
            k -= f
        raise IndexError("history row out of range")

    def last(self):
        for c, f in zip(reversed(self.chunks), reversed(self.filled)):
            if f:
                return c[f - 1]
        return None

    def reset(self):
        self.chunks, self.filled, self.caps = [], [], []
        self.version += 1


class HistoryView:
    """Read-only mapping over lazily materialised history arrays, cached until
    the owner steps again.  `view[key]` is a NumPy array (time first)."""

    def __init__(self, keys, materialise, version):
        self._keys = tuple(keys)
        self._materialise = materialise
        self._version = version
        self._cache = None
        self._cache_version = None

    def _get(self):
        v = self._version()  # (owners publish pending step-plan rows inside this call, before the version is read)
        if self._cache is None or self._cache_version != v:
            self._cache = self._materialise()
            self._cache_version = self._version()
        return self._cache

    def __getitem__(self, key):
        return self._get()[key]

    def keys(self):
        return self._keys

    def items(self):
        return self._get().items()

    def __iter__(self):
        return iter(self._keys)

    def __contains__(self, key):
        return key in self._keys

    def __len__(self):
        return len(self._keys)
