"""Running statistics of the logged scalars (interface of the reference's utils/AverageMeter.py: ``update`` / ``val`` /
``count`` / ``avg``; a meter built with a list of names tracks one channel per name and answers with lists)."""


class _Channel:
    __slots__ = ("last", "total", "n")

    def __init__(self):
        self.last, self.total, self.n = 0, 0, 0

    def push(self, v):
        self.last = v
        self.total += v
        self.n += 1

    @property
    def mean(self):
        return self.total / self.n


class AverageMeter:
    def __init__(self, items=None):
        self.items = items
        self.reset()

    @property
    def n_items(self):
        return len(self._ch)

    def reset(self):
        self._ch = [_Channel() for _ in (self.items if self.items is not None else (None,))]

    def update(self, values):
        """a list feeds the channels in order; a bare scalar feeds channel 0"""
        if isinstance(values, list):
            for ch, v in zip(self._ch, values):
                ch.push(v)
        else:
            self._ch[0].push(values)

    def _read(self, attr, idx):
        if idx is not None:
            return getattr(self._ch[idx], attr)
        out = [getattr(ch, attr) for ch in self._ch]
        return out[0] if self.items is None else out

    def val(self, idx=None):
        return self._read("last", idx)

    def count(self, idx=None):
        return self._read("n", idx)

    def avg(self, idx=None):
        return self._read("mean", idx)
