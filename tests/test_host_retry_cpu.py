"""CPU: the host side's THJ_ERETRY protocol without a device -- a stub in place of the library.  thj_fusion_finish answers THJ_ERETRY when a
single batch had more raw fusion candidates than the buffer held (the buffer has then been enlarged): Context.fusions / fusion_search make
the same calls again, at most three times, and any other status is an error at once."""
import pytest

from tophat_amd import host


class StubLib:
    def __init__(self, finish_codes):
        self.finish_codes = list(finish_codes)
        self.calls = []

    def thj_last_error(self):
        return b"stub: run the pass again"

    def thj_fusion_reset_async(self, ctx):
        self.calls.append("reset"); return 0

    def thj_fusion_set_ignored(self, ctx, p, n):
        self.calls.append("ignored"); return 0

    def thj_fusion_run_async(self, ctx, p, b):
        self.calls.append("run"); return 0

    def thj_fusion_finish(self, ctx, n):
        self.calls.append("finish")
        n._obj.value = 7
        return self.finish_codes.pop(0)


class StubParams:
    def as_ctypes(self):
        import ctypes as C
        return C.c_int(0)


class StubCtx:
    _fusion_pass = host.Context._fusion_pass

    def __init__(self, lib):
        self.lib, self._ctx = lib, None


def test_fusion_pass_runs_again_on_eretry():
    import ctypes as C
    runs = [(StubParams(), C.c_int(1)), (StubParams(), C.c_int(2))]
    lib = StubLib([-7, 0])
    assert StubCtx(lib)._fusion_pass(runs, ()) == 7
    assert lib.calls == ["reset", "ignored", "run", "run", "finish"] * 2
    lib = StubLib([0])
    assert StubCtx(lib)._fusion_pass(iter(runs), ()) == 7 and lib.calls.count("run") == 2        # (a generator of runs is kept for the rerun)
    lib = StubLib([-7, -7, -7])
    with pytest.raises(host.ThjError, match="run the pass again"):
        StubCtx(lib)._fusion_pass(runs, ())
    assert lib.calls.count("finish") == 3
    lib = StubLib([-4])
    with pytest.raises(host.ThjError):
        StubCtx(lib)._fusion_pass(runs, ())
    assert lib.calls.count("finish") == 1
