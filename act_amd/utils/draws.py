"""Keyed table of random draws (mask, DropPath gates, gumbel noise, prompt dropout).

Production runs pass ``draws=None`` and every consumer samples on the device.  Parity tests
record the draws of one run and replay them into another (CPU oracle <-> HIP path), which is
how SURVEY Appendix B's "all random draws must be injectable" is met."""


class Draws:
    def __init__(self, table=None, record=False, device=None):
        self.table = dict(table) if table else {}
        self.record = record
        self.device = device

    def get(self, key, make):
        if key in self.table:
            t = self.table[key]
            if self.device is not None and hasattr(t, "to"):
                t = t.to(self.device)
            return t
        t = make()
        if self.record:
            self.table[key] = t
        return t

    def has(self, key):
        return key in self.table
