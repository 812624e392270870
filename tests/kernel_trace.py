"""Which kernel family did the library launch?  ``with kernel_trace() as t: ...; t.counts`` is {family name: launches} from the library's
launch trace (dcpt_trace_enable / dcpt_trace_read, include/dcpt_hip.h ABI 15): every dispatch decision names what it launched, so a parity
test can assert that the result it compared came from the kernel it means to test -- a threshold edit that silently re-routes a shape
fails the assertion instead of leaving the product's kernel untested."""
import contextlib
import ctypes


class _Trace:
    def __init__(self):
        self.counts = {}

    def __getitem__(self, name):
        return self.counts.get(name, 0)

    def families(self, prefix=""):
        return {k: v for k, v in self.counts.items() if k.startswith(prefix)}

    def assert_ran(self, *names):
        for n in names:
            assert self.counts.get(n, 0) > 0, f"kernel family {n!r} did not run; launched: {self.counts}"

    def assert_not_ran(self, *names):
        for n in names:
            assert self.counts.get(n, 0) == 0, f"kernel family {n!r} ran {self.counts[n]} time(s); launched: {self.counts}"


@contextlib.contextmanager
def kernel_trace():
    from dcpt_amd import _lib

    lib = _lib.load()
    t = _Trace()
    lib.dcpt_trace_enable(1)
    try:
        yield t
    finally:
        need = lib.dcpt_trace_read(None, 0)
        buf = ctypes.create_string_buffer(int(need) + 16)
        lib.dcpt_trace_read(buf, len(buf))
        lib.dcpt_trace_enable(0)
        for line in buf.value.decode().splitlines():
            name, _, cnt = line.rpartition(" ")
            if name:
                t.counts[name] = t.counts.get(name, 0) + int(cnt)
